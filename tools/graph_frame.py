#!/usr/bin/env python3
"""Run ON the GPU box: does a hipGraph help the launch-bound shapes?  The reference's own frame (update + displacement at
WaveResolution 64 + gen of a 1024 x 1024 mesh, examples/ocean/ocean.cpp:59,135,179) and BASELINE configs[1] (512^2 x 1)
as stream launches through the C ABI and as ONE captured graph replayed (torch.cuda.CUDAGraph over the handle's stream:
dt = 1/60 and the header are the same every frame in the bench, so a replay is the same work; an integrator with a moving
camera would update the gen node's parameters per frame on top of this)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from datum_amd import capi, host_api

DT = np.float32(1 / 60)
dev = torch.device("cuda", 0)


def timed(stream, fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        fn()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(N, mesh, label, unroll):
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    with capi.Ocean(N, 1) as oc:
        stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(stream)
        oc.set_stream(stream.cuda_stream)
        oc.set_cascade(0, host_api.EXAMPLE_TUNABLES["wavescale"], 1.35)
        oc.upload_state(0, p.height)
        header = p.oceanset()
        verts = torch.empty(max(mesh, 2) * max(mesh, 2) * 12, dtype=torch.float32, device=dev)

        def frame():
            oc.update(DT)
            oc.displace()
            if mesh:
                oc.gen(0, header, mesh, mesh, verts.data_ptr())

        for _ in range(10):
            frame()
        torch.cuda.synchronize()
        plain = timed(stream, frame, 400)
        res = [f"{label}: stream launches {plain:7.2f} us"]
        for k in unroll:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for _ in range(k):
                        frame()
                torch.cuda.synchronize()
                t = timed(stream, g.replay, 400 // k) / k
                res.append(f"graph of {k} frame(s) {t:7.2f} us/frame")
            except Exception as e:  # noqa: BLE001
                res.append(f"graph of {k}: capture failed: {str(e).splitlines()[0][:120]}")
                torch.cuda.synchronize()
        print(";  ".join(res), flush=True)
        oc.set_stream(None)
        torch.cuda.set_stream(torch.cuda.default_stream(dev))


case(64, 1024, "reference frame (64^2 + 1024^2 mesh)", (1, 4))
case(64, 0, "64^2 displacement only", (1, 4))
case(512, 0, "512^2 x 1 step (configs[1])", (1, 4))
case(1024, 0, "1024^2 x 1 step", (1,))

#!/usr/bin/env python3
"""ocean.gen timing on the GPU box: 1024 x 1024 mesh (examples/ocean/ocean.cpp:59) from maps of several resolutions.
usage: python tools/gen_bench.py [N ...]     prints one line per N (event time per launch, bytes / time against 8 TB/s)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from datum_amd import capi, host_api

sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1024]
sx = sy = 1024
for N in sizes:
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    with capi.Ocean(N, 1) as oc:
        stream = torch.cuda.Stream()
        oc.set_stream(stream.cuda_stream)
        oc.set_cascade(0, host_api.EXAMPLE_TUNABLES["wavescale"], 1.35)
        oc.upload_state(0, p.height)
        for _ in range(10):
            oc.update(np.float32(1 / 60))
        oc.displace()
        gs = p.oceanset()
        verts = torch.empty(sx * sy * 12, dtype=torch.float32, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20):
            oc.gen(0, gs, sx, sy, verts.data_ptr())
        e0.record(stream)
        reps = 200
        for _ in range(reps):
            oc.gen(0, gs, sx, sy, verts.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        nbytes = 48.0 * sx * sy + 32.0 * N * N
        print(f"gen {sx}x{sy} mesh from {N:4d}^2 maps: {us:7.2f} us  {sx * sy / us * 1e-3:6.2f} G vertices/s  "
              f"{nbytes / us * 1e-6:6.2f} TB/s on {nbytes / 1e6:.1f} MB algorithmic = {nbytes / us * 1e-6 / 8.0:.3f} of 8 TB/s", flush=True)
        oc.set_stream(None)

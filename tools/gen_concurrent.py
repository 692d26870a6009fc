#!/usr/bin/env python3
"""Run ON the GPU box: does ocean.gen overlap with itself?  One launch of a 1024 x 1024 mesh against two launches of
1024 x 512 meshes (two handles, same map contents) on two streams at the same time, and against the two back to back."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from datum_amd import capi, host_api

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
p.seed_ocean(1000)
gs = p.oceanset()
ocs, streams, verts = [], [], []
for k in range(2):
    oc = capi.Ocean(N, 1)
    st = torch.cuda.Stream()
    oc.set_stream(st.cuda_stream)
    oc.set_cascade(0, 22.0, 1.35)
    oc.upload_state(0, p.height)
    oc.update(np.float32(1 / 60))
    oc.displace()
    ocs.append(oc); streams.append(st); verts.append(torch.empty(1024 * 1024 * 12, dtype=torch.float32, device="cuda"))
torch.cuda.synchronize()


def timed(fn, reps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0.record(streams[0])
    streams[1].wait_event(e0)
    for _ in range(reps):
        fn()
    e = torch.cuda.Event(); e.record(streams[1]); streams[0].wait_event(e)
    e1.record(streams[0])
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


one = timed(lambda: ocs[0].gen(0, gs, 1024, 1024, verts[0].data_ptr()))
half = timed(lambda: ocs[0].gen(0, gs, 1024, 512, verts[0].data_ptr()))
seq = timed(lambda: (ocs[0].gen(0, gs, 1024, 512, verts[0].data_ptr()), ocs[0].gen(0, gs, 1024, 512, verts[1].data_ptr())))
par = timed(lambda: (ocs[0].gen(0, gs, 1024, 512, verts[0].data_ptr()), ocs[1].gen(0, gs, 1024, 512, verts[1].data_ptr())))
par4 = timed(lambda: (ocs[0].gen(0, gs, 1024, 256, verts[0].data_ptr()), ocs[1].gen(0, gs, 1024, 256, verts[1].data_ptr()),
                      ocs[0].gen(0, gs, 1024, 256, verts[0].data_ptr()), ocs[1].gen(0, gs, 1024, 256, verts[1].data_ptr())))
print(f"N={N}: one 1024x1024 launch {one:.2f} us; one 1024x512 launch {half:.2f}; two 1024x512 back to back on one stream {seq:.2f}; "
      f"two 1024x512 on two streams at once {par:.2f}; four 1024x256 on two streams {par4:.2f}")
for oc in ocs:
    oc.set_stream(None)

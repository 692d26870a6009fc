"""Run ON the GPU box: would the step gain from kernels of different steps overlapping?  The same 4 grids of 1024^2 as ONE handle of four cascades on
one stream (the shipped shape: row pass, column pass, row pass, ... strictly one after the other) and as TWO handles of two cascades on two streams
(a row pass of one may run beside a column pass of the other: the tails of one kernel under the body of the next)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datum_amd import capi, farm, host_api          # noqa: E402

N = int(os.environ.get("N", "1024"))
STEPS = int(os.environ.get("STEPS", "2000"))
DT = 1.0 / 60


def seeded(C, first):
    oc = capi.Ocean(N, C)
    for c in range(C):
        ws = farm.grid_wavescale(first + c, 4)
        p = host_api.OceanParams(N, **dict(host_api.EXAMPLE_TUNABLES, wavescale=ws))
        p.seed_ocean(farm.grid_seed(first + c))
        oc.set_cascade(c, ws, 1.35)
        oc.upload_state(c, p.height)
        del p
    return oc


def run(handles, streams, label):
    for oc, st in zip(handles, streams):
        oc.set_stream(st.cuda_stream)
    for _ in range(50):
        for oc in handles:
            oc.update(DT)
            oc.displace()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for oc in handles:
            oc.update(DT)
            oc.displace()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    grids = sum(oc.cascades for oc in handles)
    print(f"  {label:<64s} {dt / STEPS * 1e6:8.2f} us per step of {grids} grids   {grids * STEPS / dt:9.0f} grids/s")
    for oc in handles:
        oc.set_stream(None)


torch.zeros(1, device="cuda:0")
for rep in range(3):
    one = seeded(4, 0)
    run([one], [torch.cuda.Stream()], "one handle x 4 cascades, one stream")
    del one
    a, b = seeded(2, 0), seeded(2, 2)
    run([a, b], [torch.cuda.Stream(), torch.cuda.Stream()], "two handles x 2 cascades, two streams")
    s = torch.cuda.Stream()
    run([a, b], [s, s], "two handles x 2 cascades, one stream")
    del a, b
    hs = [seeded(1, i) for i in range(4)]
    run(hs, [torch.cuda.Stream() for _ in range(4)], "four handles x 1 cascade, four streams")
    del hs

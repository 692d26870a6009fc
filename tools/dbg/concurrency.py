"""Experiment: does running independent handles on separate streams (row pass of one overlapping the column pass of
another) beat one handle with all cascades?  Same total work: 4 grids of 1024^2 per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from datum_amd import capi, host_api
N = 1024; DT = np.float32(1/60)
def make(ncasc):
    oc = capi.Ocean(N, ncasc)
    for c in range(ncasc):
        p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES); p.seed_ocean(1000 + c)
        oc.set_cascade(c, 22.0, 1.35); oc.upload_state(c, p.height)
    return oc
def run(handles, steps=200):
    for oc in handles:
        for _ in range(10): oc.update(DT); oc.displace()
    for oc in handles: oc.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for oc in handles:
            oc.update(DT); oc.displace()
    for oc in handles: oc.sync()
    el = time.perf_counter() - t0
    return el / steps * 1e6
for split in ((4,), (2, 2), (1, 1, 1, 1)):
    hs = [make(c) for c in split]
    us = run(hs)
    print(f"handles {split}: {us:7.1f} us per step of 4 grids  -> {4/us*1e6:8.0f} grids/s")
    for h in hs: h.close()

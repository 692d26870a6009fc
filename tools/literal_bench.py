#!/usr/bin/env python3
"""Validation mode timing on the GPU box: one displacement step through the reference's own algorithm (datum_ocean_set_literal_transform)
against the fused step, per resolution.  usage: python tools/literal_bench.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: F401  (loads the HIP runtime the module links against)

from datum_amd import capi, host_api

DT = np.float32(1 / 60)
for N in [int(a) for a in sys.argv[1:]] or [64, 1024, 4096]:
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    with capi.Ocean(N, 1) as oc:
        oc.set_cascade(0, host_api.EXAMPLE_TUNABLES["wavescale"], 1.35)
        oc.upload_state(0, p.height)
        out = []
        for literal in (False, True):
            oc.set_literal_transform(literal)
            steps = 200 if not literal else 20
            for _ in range(3):
                oc.update(DT); oc.displace()
            oc.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                oc.update(DT); oc.displace()
            oc.sync()
            out.append((time.perf_counter() - t0) / steps * 1e6)
    print(f"{N:5d}^2 x 1: fused step {out[0]:9.1f} us   literal-transform step {out[1]:10.1f} us   ({out[1] / out[0]:.0f} x)")

"""Diagnostic: two OceanParams alternating on one OceanContext; after every displace the device phase (read through the C ABI
handle of the context, no fetch) is compared with the oracle's: prints the first frames at which either state is off."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from datum_amd import capi, host_api
from oracle import oracle

N = 64
dt = np.float32(1 / 60)
ws = (22.0, 64.0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4400
ps = []
for k in range(2):
    p = host_api.OceanParams(N, **dict(host_api.EXAMPLE_TUNABLES, wavescale=ws[k]))
    p.seed_ocean(1000 + k)
    ps.append(p)
phases = [np.zeros((N, N), np.float32) for _ in range(2)]
lib = capi.load()
bad = 0
with host_api.OceanContext(N, device=0) as ctx:
    h = ctypes.c_void_p(host_api.load().datum_host_context_handle(ctx.c))
    got = np.empty((N, N), np.float32)
    for frame in range(steps):
        for k in range(2):
            ps[k].update_ocean(dt)
            oracle.update(phases[k], ws[k], dt)
            ctx.displace_ocean_surface(ps[k])
            rc = lib.datum_ocean_read_state(h, 0, got.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0
            if not np.array_equal(got, phases[k]):
                bad += 1
                if bad <= 10:
                    d = float(got[0, 0]) - float(phases[k][0, 0])
                    print(f"frame {frame} state {k}: device phase differs, [0][0] by {d:+.6f} (one step at ws 22: 0.1878, at ws 64: 0.1100)", flush=True)
                phases[k][...] = got          # go on from what the device has
print("frames", steps, "mismatches", bad)

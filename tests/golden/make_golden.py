"""Regenerates tests/golden/ocean_n64.npz from the CPU oracle (oracle/ocean_oracle.cpp).

The reference holds no golden vectors for this path and cannot be built here, so these
fixtures pin the ORACLE's output at the reference's own size (WaveResolution = 64, example-ocean
parameters, examples/ocean/ocean.cpp:46-50, dt = 1/60 as examples/example-xcb.cpp:1100) against
regressions, and give the GPU tests a committed target.  Run: python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as o  # noqa: E402
import gen_cases  # noqa: E402

N = 64
SEED = 1000
DT = np.float32(1.0 / 60.0)
STEPS = (1, 60, 600)


def main():
    p = o.EXAMPLE
    seed, h0, rej = o.seed(N, SEED, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"], sanitize=False, return_rejected=True)
    assert rej == 0
    out = dict(seed=seed, h0=h0)
    phase = np.zeros((N, N), np.float32)
    w = o.weights(N)
    done = 0
    for steps in STEPS:
        for _ in range(steps - done):
            o.update(phase, p["wavescale"], DT)
        done = steps
        m = o.displace(h0, phase.copy(), p["wavescale"], p["choppiness"], dt=0.0, w=w)
        out[f"phase_{steps}"] = phase.copy()
        out[f"maps_{steps}"] = m
    s = o.example_oceanset(N, swellphase=0.25)
    out["vertices_600_32x32"] = o.gen(s, out["maps_600"], 32, 32)
    out["oceanset"] = np.frombuffer(bytes(s), np.uint8).copy()
    # a steep swell seen from a camera pitched 30 degrees down: every qi * ... term of gen.comp:97-105 is non-zero
    steep = gen_cases.oceanset(o, N, "pitched_steep", swellphase=1.9)
    out["vertices_600_steep_48x40"] = o.gen(steep, out["maps_600"], 48, 40)
    out["oceanset_steep"] = np.frombuffer(bytes(steep), np.uint8).copy()
    # a rolled camera over a swell running along y, and a sea level off z = 0 seen from a camera pitched 20 degrees down
    for name in ("rolled", "plane_w"):
        hdr = gen_cases.oceanset(o, N, name, swellphase=0.9)
        out[f"vertices_600_{name}_40x30"] = o.gen(hdr, out["maps_600"], 40, 30)
        out[f"oceanset_{name}"] = np.frombuffer(bytes(hdr), np.uint8).copy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ocean_n64.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()

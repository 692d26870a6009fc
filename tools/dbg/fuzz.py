"""Diagnostic, run ON the GPU box: seeded random differential test of the HIP path against the oracle -- resolutions, cascade counts,
wave scales, choppiness, amplitudes, runs of updates (some negative / large), then ocean.gen under a random camera case, mesh size
(ragged, sometimes > 1024 along one axis) and cascade.  Prints one line per case and a summary; exit code 1 on the first mismatch.
usage: python tools/dbg/fuzz.py [cases=60] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch

import gen_cases
from datum_amd import capi
from oracle import oracle

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
p = oracle.EXAMPLE
names = list(gen_cases.CASES)
ONLY = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None      # run this one case (the draws of the others are consumed, nothing else)


def rmse(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).mean()))


worst = dict(maps=0.0, normal=0.0, pos=0.0, frame=0.0)
for k in range(cases):
    N = int(rng.choice([64, 128, 256, 512, 1024], p=[0.25, 0.25, 0.25, 0.15, 0.10])) if not os.environ.get("FUZZ_SIZES") else int(rng.choice([int(v) for v in os.environ["FUZZ_SIZES"].split(",")]))
    C = int(rng.integers(1, 4)) if N <= 1024 else int(rng.integers(1, 3))
    half = bool(rng.random() < 0.3)
    h0half = half and bool(rng.random() < 0.5)          # DATUM_OCEAN_SPECTRUM_FP16_H0: h0 read as halves too
    scales = np.exp(rng.uniform(np.log(2.0), np.log(600.0), C)).astype(np.float32)
    chops = rng.uniform(0.0, 2.0, C).astype(np.float32)
    amps = (0.0025 * 10.0 ** rng.uniform(-1.5, 1.5, C)).astype(np.float32)
    # every random draw of the case up front, so that FUZZ_ONLY=k can skip the other cases' GPU and oracle work and still arrive at case k's draws
    runs = [[np.float32(rng.choice([1 / 60, 1 / 30, 0.25, -1 / 60, 3.5, 0.0], p=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05])) for _ in range(int(rng.integers(1, 11)))] for rnd in range(int(rng.integers(1, 4)))]
    c_gen = int(rng.integers(0, C))
    case = names[int(rng.integers(0, len(names)))]
    sx, sy = int(rng.integers(2, 220)), int(rng.integers(2, 220))
    if rng.random() < 0.15:
        sx, sy = (int(rng.integers(1025, 1400)), int(rng.integers(2, 40))) if rng.random() < 0.5 else (int(rng.integers(2, 40)), int(rng.integers(1025, 1400)))
    swellphase = float(rng.uniform(0, 6.28))
    if ONLY is not None and k != ONLY:
        continue
    states = [oracle.seed(N, 3000 + 7 * k + c, float(scales[c]), float(amps[c]), p["windspeed"], p["winddirection"], sanitize=True)[1] for c in range(C)]
    phases = [np.zeros((N, N), np.float32) for _ in range(C)]
    w = oracle.weights(N, reduced=True)
    with capi.Ocean(N, C) as oc:
        oc.set_spectrum_format("fp16h0" if h0half else half)
        for c in range(C):
            oc.set_cascade(c, float(scales[c]), float(chops[c]))
            oc.upload_state(c, states[c])
        for run in runs:
            for dt in run:
                oc.update(float(dt))
                for c in range(C):
                    oracle.update(phases[c], float(scales[c]), dt)
            oc.displace()
        maps = []
        for c in range(C):
            assert np.array_equal(oc.read_state(c), phases[c]), (k, "phase", c)
            ref = oracle.displace(states[c], phases[c].copy(), float(scales[c]), float(chops[c]), w=w, mt=N > 1024)
            got = oc.read_maps(c)
            big = max(float(np.abs(ref[0]).max()), 1e-30)
            e = rmse(got[0][..., :3], ref[0][..., :3]) / big
            en = float(np.abs(got[1][..., :3] - ref[1][..., :3]).max())
            assert np.isfinite(got).all() and np.all(got[..., 3] == 0)
            assert e < (2e-3 if half else 2e-6), (k, "maps", c, e)
            # a unit normal is normalize(height differences, 4 / (scale N)): an absolute error d in the heights moves it by up to ~ 2 d / nz
            nzterm = 4.0 / ((1.0 / float(scales[c])) * N)
            allowed = (2e-2 if half else 2e-5) + 8.0 * e * big / nzterm
            assert en < allowed, (k, "normal", c, en, allowed)
            if not half:
                worst["maps"], worst["normal"] = max(worst["maps"], e), max(worst["normal"], en)
            maps.append(got)
        # ocean.gen from one of the cascades
        c = c_gen
        s = gen_cases.oceanset(oracle, N, case, swellphase=swellphase, wavescale=float(scales[c]))
        s.choppiness = float(chops[c])
        verts = torch.full((sx * sy * 12 + 64,), 777.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        oc.gen(c, capi.OceanSet.from_buffer_copy(bytes(s)), sx, sy, verts.data_ptr())
        oc.sync()
        torch.cuda.synchronize()
        v = verts.cpu().numpy()
        assert np.all(v[-64:] == 777.0), (k, "gen wrote past the mesh")
        got = v[:-64].reshape(sy, sx, 12)
        want = oracle.gen(s, maps[c], sx, sy)
        pos, tex, frame = gen_cases.compare(got, want)
        # Far vertices (round 6, seed 8100 case 343: a vertex 980 km out, texel coordinate 4.4e6 -- ONE fractional bit left in fp32): the two evaluations'
        # texture coordinates differ by an ulp, i.e. by half a texel, and the sampled displacement by the map's own variation there (1e-3 m; against
        # float64 both are 6e-3 off).  The position bar therefore grows by what the map can change over the distance between the two evaluations' OWN
        # sample points (their texture coordinates are both in the vertex): |texel offset| x the largest jump between neighbouring texels (x 2: both
        # axes).  Where the coordinates are the same floats -- nearly everywhere -- the plain 2e-4 stands.
        m0 = maps[c][0][..., :3].astype(np.float64)
        jump = max(float(np.abs(np.roll(m0, 1, 0) - m0).max()), float(np.abs(np.roll(m0, 1, 1) - m0).max()))
        texels = np.abs(got[..., 3:5].astype(np.float64) - want[..., 3:5]).max(-1) * N
        allow = 2e-4 * (1 + np.abs(want[..., 0:3].astype(np.float64))) + 2.0 * jump * texels[..., None]
        pos = float((np.abs(got[..., 0:3].astype(np.float64) - want[..., 0:3]) / allow).max()) * 2e-4
        if os.environ.get("FUZZ_DUMP"):
            np.savez(os.environ["FUZZ_DUMP"], got=got, want=want, maps=maps[c], oceanset=np.frombuffer(bytes(s), np.uint8), N=N, sx=sx, sy=sy, scale=float(scales[c]), chop=float(chops[c]), swellphase=swellphase)
        assert np.isfinite(got).all() and np.all(got[..., 11] == -1)
        assert pos < 2e-4 and tex < 2e-4 and frame < 2e-4, (k, "gen", case, N, sx, sy, pos, tex, frame)
        worst["pos"], worst["frame"] = max(worst["pos"], pos), max(worst["frame"], frame)
    print(f"case {k:3d}: N={N:4d} x {C} {('fp16h0' if h0half else 'fp16') if half else 'fp32'} scales {[round(float(x), 1) for x in scales]}  gen {case} cascade {c} mesh {sx}x{sy}: ok", flush=True)
print(f"{cases} cases ok; worst fp32 displacement rmse / max {worst['maps']:.2e}, normal max abs {worst['normal']:.2e}, vertex position {worst['pos']:.2e}, frame {worst['frame']:.2e}")

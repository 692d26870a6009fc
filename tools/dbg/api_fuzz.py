"""Diagnostic, run ON the GPU box: seeded random SEQUENCES of C-ABI calls on one handle against a host model of what the module should
hold -- updates (queued, some negative / large / zero), displacements, wave-scale and choppiness changes with updates pending, toggles of the
spectrum format and of the literal-transform validation mode, state uploads with and without a phase, reads of the state, park / resume through caller-owned device memory (also into
another cascade), device-side rebuilds of h0 from the seed, maps bound to caller memory and back, changes of stream, ocean.gen.  After every
displacement each cascade's maps are compared with the oracle on the model's state; every read of the state is compared bit for bit.
usage: python tools/dbg/api_fuzz.py [sequences=40] [seed=1] [ops per sequence=40]      (FUZZ_SIZES=1024,2048: those resolutions instead of 64 ... 512)"""
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
sequences = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nops = int(sys.argv[3]) if len(sys.argv) > 3 else 40
p = oracle.EXAMPLE
names = list(gen_cases.CASES)


def rmse(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).mean()))


counts = {}
worst = dict(maps=0.0, normal=0.0, pos=0.0)

for q in range(sequences):
    N = int(rng.choice([64, 128, 256, 512], p=[0.35, 0.3, 0.25, 0.1])) if not os.environ.get("FUZZ_SIZES") else int(rng.choice([int(v) for v in os.environ["FUZZ_SIZES"].split(",")]))
    C = int(rng.integers(1, 5)) if N <= 512 else int(rng.integers(1, 3))
    w = oracle.weights(N, reduced=True)
    wlit = oracle.weights(N)          # the reference's literal table: what the literal-transform mode is compared with
    log = []

    def new_state(tag):
        scale = float(np.exp(rng.uniform(np.log(4.0), np.log(400.0))))
        amp = float(0.0025 * 10.0 ** rng.uniform(-1.0, 1.0))
        seed, h0 = oracle.seed(N, 9000 + 131 * q + tag, scale, amp, p["windspeed"], p["winddirection"], sanitize=True)
        return seed, h0

    # the model: what every cascade should hold
    scale = [float(np.exp(rng.uniform(np.log(4.0), np.log(400.0)))) for _ in range(C)]
    chop = [float(rng.uniform(0.0, 2.0)) for _ in range(C)]
    h0, phase, seeds = [], [], []
    half = False
    fmt = "fp32"
    literal = False
    parked = []           # (device tensor, flags, h0, phase)
    displaced = False     # the maps are those of the model's state
    bound = None          # caller-owned maps tensor (or None: the handle's own)
    streams = [None, torch.cuda.Stream(), torch.cuda.Stream()]

    with capi.Ocean(N, C) as oc:
        for c in range(C):
            s, h = new_state(c)
            seeds.append(None)
            h0.append(h)
            phase.append(np.zeros((N, N), np.float32))
            oc.set_cascade(c, scale[c], chop[c])
            oc.upload_state(c, h)

        def check_maps():
            for c in range(C):
                ref = oracle.displace(h0[c], phase[c].copy(), scale[c], chop[c], w=wlit if literal else w)
                got = oc.read_maps(c)
                big = max(float(np.abs(ref[0]).max()), 1e-30)
                e = rmse(got[0][..., :3], ref[0][..., :3]) / big
                en = float(np.abs(got[1][..., :3] - ref[1][..., :3]).max())
                assert np.isfinite(got).all() and np.all(got[..., 3] == 0), (q, log[-12:])
                coarse = half                       # (never together with the literal mode: the two refuse each other)
                assert e < (2e-3 if coarse else 2e-6), (q, "maps", c, e, half, literal, log[-12:])
                nzterm = 4.0 / ((1.0 / scale[c]) * N)
                # (a slope is the difference of two heights over a vector at least nzterm long: the normal may be off by the heights' error / nzterm)
                emax = float(np.abs(got[0][..., 2].astype(np.float64) - ref[0][..., 2]).max())
                allowed = (2e-2 if coarse else 2e-5) + max(8.0 * e * big, 4.0 * emax) / nzterm
                assert en < allowed, (q, "normal", c, en, allowed, log[-12:])
                if not coarse:
                    worst["maps"], worst["normal"] = max(worst["maps"], e), max(worst["normal"], en)

        ops = ["update", "displace", "set_cascade", "format", "read_state", "upload", "park", "resume", "rebuild", "bind", "stream", "gen", "literal"]
        weights = np.array([0.26, 0.2, 0.08, 0.05, 0.07, 0.06, 0.06, 0.06, 0.04, 0.04, 0.04, 0.04, 0.04])

        for k in range(nops):
            op = str(rng.choice(ops, p=weights / weights.sum()))
            counts[op] = counts.get(op, 0) + 1

            if op == "update":
                dt = np.float32(rng.choice([1 / 60, 1 / 30, 0.25, -1 / 60, 3.5, 0.0], p=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05]))
                log.append(("update", float(dt)))
                oc.update(float(dt))
                for c in range(C):
                    oracle.update(phase[c], scale[c], dt)
                displaced = False
            elif op == "displace":
                log.append(("displace",))
                oc.displace()
                check_maps()
                displaced = True
            elif op == "set_cascade":
                c = int(rng.integers(0, C))
                if rng.random() < 0.6:
                    scale[c] = float(np.exp(rng.uniform(np.log(4.0), np.log(400.0))))
                chop[c] = float(rng.uniform(0.0, 2.0))
                log.append(("set_cascade", c, scale[c], chop[c]))
                oc.set_cascade(c, scale[c], chop[c])      # (updates queued before it were issued under the old wave scale: the model applied them at once)
                displaced = False
            elif op == "format":
                # (ABI 7: the fp16 format and the literal mode -- the reference's fp32 arithmetic -- refuse each other with ESTATE instead of one silently winning)
                if literal and not half:
                    try:
                        oc.set_spectrum_format(str(rng.choice(["fp16", "fp16h0"])))
                        raise AssertionError((q, "fp16 format accepted in literal mode", log[-12:]))
                    except capi.OceanError as e:
                        assert e.code == capi.ESTATE
                    log.append(("format refused", half))
                else:
                    # (fp32 <-> one of the two fp16 formats, or from one fp16 format to the other: h0 read as halves too, DATUM_OCEAN_SPECTRUM_FP16_H0)
                    fmt = str(rng.choice([f for f in ("fp32", "fp16", "fp16h0") if f != fmt]))
                    half = fmt != "fp32"
                    log.append(("format", fmt))
                    oc.set_spectrum_format(fmt)
                displaced = False
            elif op == "literal":
                if half and not literal:
                    try:
                        oc.set_literal_transform(True)
                        raise AssertionError((q, "literal mode accepted with the fp16 spectrum", log[-12:]))
                    except capi.OceanError as e:
                        assert e.code == capi.ESTATE
                    log.append(("literal refused", literal))
                else:
                    literal = not literal
                    log.append(("literal", literal))
                    oc.set_literal_transform(literal)
                displaced = False
            elif op == "read_state":
                c = int(rng.integers(0, C))
                log.append(("read_state", c))
                assert np.array_equal(oc.read_state(c), phase[c]), (q, "phase", c, log[-12:])
            elif op == "upload":
                c = int(rng.integers(0, C))
                _, h = new_state(1000 + k)
                withphase = rng.random() < 0.5
                ph = rng.uniform(-20.0, 20.0, (N, N)).astype(np.float32) if (withphase and rng.random() < 0.3) else rng.uniform(0.0, 6.2831, (N, N)).astype(np.float32)
                log.append(("upload", c, withphase))
                oc.upload_state(c, h, ph if withphase else None)
                h0[c] = h
                phase[c] = ph.copy() if withphase else np.zeros((N, N), np.float32)
                seeds[c] = None
                displaced = False
            elif op == "park":
                c = int(rng.integers(0, C))
                buf = torch.empty(oc.state_bytes() // 4, dtype=torch.float32, device="cuda:0")
                flags = oc.park_state(c, buf.data_ptr(), oc.state_bytes())
                log.append(("park", c, flags))
                parked.append((buf, flags, h0[c].copy(), phase[c].copy()))
                if len(parked) > 4:
                    parked.pop(0)
            elif op == "resume":
                if not parked:
                    continue
                c = int(rng.integers(0, C))
                buf, flags, h, ph = parked[int(rng.integers(0, len(parked)))]
                log.append(("resume", c, flags))
                oc.resume_state(c, buf.data_ptr(), oc.state_bytes(), flags)
                h0[c], phase[c] = h.copy(), ph.copy()
                seeds[c] = None
                displaced = False
            elif op == "rebuild":
                c = int(rng.integers(0, C))
                if seeds[c] is None or rng.random() < 0.3:
                    seeds[c], _ = new_state(2000 + k)
                    oc.upload_seed(c, seeds[c])
                sc = float(np.exp(rng.uniform(np.log(4.0), np.log(400.0))))
                amp = float(0.0025 * 10.0 ** rng.uniform(-1.0, 1.0))
                wind = float(rng.uniform(3.0, 15.0))
                ang = float(rng.uniform(0, 6.28))
                wd = (float(np.float32(np.cos(ang))), float(np.float32(np.sin(ang))))
                log.append(("rebuild", c, sc, amp, wind, wd))
                oc.rebuild_height(c, sc, amp, wind, wd)
                want = oracle.height_from_seed(seeds[c], sc, amp, wind, wd)
                got = oc.read_height(c)
                assert np.abs(got - want).max() <= 2e-6 * max(float(np.abs(want).max()), 1e-30), (q, "rebuild", c, log[-12:])
                h0[c] = got            # (expf / division differ from libm by ulps: the model goes on with the device's h0)
                scale[c] = sc
                displaced = False
            elif op == "bind":
                if bound is None:
                    bound = torch.full((C * capi.map_block_floats(N) + 64,), 555.0, dtype=torch.float32, device="cuda:0")
                    log.append(("bind", "caller"))
                    oc.bind_maps(bound.data_ptr(), C * capi.map_block_floats(N) * 4)
                else:
                    oc.sync()
                    torch.cuda.synchronize()
                    assert bool((bound[-64:] == 555.0).all()), (q, "wrote past the bound maps", log[-12:])
                    log.append(("bind", "own"))
                    oc.bind_maps(0, 0)
                    bound = None
                displaced = False
            elif op == "stream":
                s = streams[int(rng.integers(0, len(streams)))]
                log.append(("stream", None if s is None else "torch"))
                oc.sync()
                torch.cuda.synchronize()
                oc.set_stream(None if s is None else s.cuda_stream)
            elif op == "gen":
                if not displaced:
                    oc.displace()
                    check_maps()
                    displaced = True
                c = int(rng.integers(0, C))
                case = names[int(rng.integers(0, len(names)))]
                sx, sy = int(rng.integers(2, 120)), int(rng.integers(2, 120))
                s = gen_cases.oceanset(oracle, N, case, swellphase=float(rng.uniform(0, 6.28)), wavescale=scale[c])
                s.choppiness = chop[c]
                log.append(("gen", c, case, sx, sy))
                verts = torch.full((sx * sy * 12 + 64,), 777.0, dtype=torch.float32, device="cuda:0")
                torch.cuda.synchronize()
                maps = oc.read_maps(c)
                oc.gen(c, capi.OceanSet.from_buffer_copy(bytes(s)), sx, sy, verts.data_ptr())
                oc.sync()
                torch.cuda.synchronize()
                v = verts.cpu().numpy()
                assert np.all(v[-64:] == 777.0), (q, "gen wrote past the mesh", log[-12:])
                got = v[:-64].reshape(sy, sx, 12)
                want = oracle.gen(s, maps, sx, sy)
                pos, tex, frame = gen_cases.compare(got, want)
                assert np.isfinite(got).all() and np.all(got[..., 11] == -1), (q, log[-12:])
                # The sequences hold seas that no wave scale was made for (h0 of one scale under another, ten times the amplitude): metres
                # of displacement from one texel to the next.  A vertex far out samples the maps at 1e4 ... 1e5 texels, where a float
                # resolves 1e-3 ... 1e-2 of a texel, and the last bits of the swell's sin / cos decide which: the bar grows with the
                # roughness of the maps (a WRONG texel would be off by the roughness itself, 50 times the allowance).
                rough = max(float(np.abs(np.diff(maps[0][..., :3], axis=0)).max()), float(np.abs(np.diff(maps[0][..., :3], axis=1)).max()))
                roughn = max(float(np.abs(np.diff(maps[1][..., :3], axis=0)).max()), float(np.abs(np.diff(maps[1][..., :3], axis=1)).max()))
                assert pos < 2e-4 + 2e-2 * rough and tex < 2e-4 and frame < 2e-4 + 2e-2 * roughn, (q, "gen", case, N, sx, sy, pos, tex, frame, rough, roughn, log[-12:])
                if rough < 1e-2:
                    worst["pos"] = max(worst["pos"], pos)

        # at the end of the sequence everything once more
        oc.displace()
        check_maps()
        for c in range(C):
            assert np.array_equal(oc.read_state(c), phase[c]), (q, "phase at the end", c, log[-12:])
        oc.sync()
        torch.cuda.synchronize()
        if bound is not None:
            assert bool((bound[-64:] == 555.0).all())
            oc.bind_maps(0, 0)
        oc.set_stream(None)
    print(f"sequence {q:3d}: N={N:4d} x {C}, {len(log)} calls: ok", flush=True)

print(f"{sequences} sequences ok; calls by kind {dict(sorted(counts.items()))}; worst fp32 displacement rmse / max {worst['maps']:.2e}, normal max abs {worst['normal']:.2e}, vertex position (maps smoother than 1 cm per texel) {worst['pos']:.2e}")

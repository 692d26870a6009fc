"""Diagnostic, run ON the GPU box: seeded random sequences of calls on the C++ host shim (datum's ocean API: datum_amd/host/, through
host_api.py) against a host model -- several OceanParams (seeded, re-seeded, copied and diverged, some with hostphase), one or two
OceanContexts, update_ocean (negative / large / zero steps), lerp_ocean_waves with steps pending (the dispersion of the OLD wave scale for
the steps issued before it), displacement of any params on any context in any order (more states than a context parks), fetch_ocean_state,
release_parked_states.  After every displacement the context's maps are compared with the oracle on the model's state of that params;
every fetched (or host-advanced) phase is compared bit for bit.
usage: python tools/dbg/host_fuzz.py [sequences=30] [seed=1] [calls per sequence=80]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from datum_amd import host_api
from oracle import oracle

oracle.build()
sequences = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ncalls = int(sys.argv[3]) if len(sys.argv) > 3 else 80


def rmse(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).mean()))


class Model:
    """what one OceanParams should be"""

    def __init__(self, p, phase, hostphase=False):
        self.p = p
        self.phase = phase
        self.hostphase = hostphase
        self.unfetched = 0        # update_ocean calls since the host phase was last current

    @property
    def wavescale(self):
        return float(self.p.scalars().wavescale)

    @property
    def choppiness(self):
        return float(self.p.scalars().choppiness)


counts = {}
worst = 0.0

for q in range(sequences):
    N = int(rng.choice([64, 128], p=[0.6, 0.4]))
    w = oracle.weights(N, reduced=True)
    log = []
    contexts = [host_api.OceanContext(N) for _ in range(int(rng.integers(1, 3)))]
    objs = []

    def new_params(k):
        tun = dict(host_api.EXAMPLE_TUNABLES, wavescale=float(np.exp(rng.uniform(np.log(6.0), np.log(300.0)))), waveamplitude=float(0.0025 * 10.0 ** rng.uniform(-0.7, 0.7)))
        p = host_api.OceanParams(N, **tun)
        hp = bool(rng.random() < 0.25)
        p.set_hostphase(hp)             # (explicitly: on by default at 64 x 64 only)
        p.seed_ocean(5000 + 97 * q + k)
        return Model(p, np.zeros((N, N), np.float32), hp)

    for i in range(int(rng.integers(2, 5))):
        objs.append(new_params(i))

    def check_phase(m, where):
        assert np.array_equal(np.asarray(m.p.phase), m.phase), (q, where, log[-14:])

    try:
        for k in range(ncalls):
            op = str(rng.choice(["update", "displace", "lerp", "fetch", "copy", "reseed", "release", "new", "burst"], p=[0.38, 0.3, 0.07, 0.08, 0.05, 0.03, 0.04, 0.03, 0.02]))
            counts[op] = counts.get(op, 0) + 1
            i = int(rng.integers(0, len(objs)))
            m = objs[i]

            if op == "update":
                dt = np.float32(rng.choice([1 / 60, 1 / 30, 0.25, -1 / 60, 3.5, 0.0], p=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05]))
                log.append(("update", i, float(dt)))
                m.p.update_ocean(float(dt))
                oracle.update(m.phase, m.wavescale, dt)
                m.unfetched += 1
                if m.hostphase:
                    check_phase(m, "hostphase after update")
            elif op == "burst":
                # more steps than the history records (OceanParams::MaxRecordedUpdates = 4096) without a render in between, now and then
                # (a params without hostphase has to be fetched -- or rendered by a context that still holds its copy -- within
                # MaxRecordedUpdates / 2 steps: documented, datum_amd/host/ocean.h; the sequences keep to it and fetch first)
                n = int(rng.choice([300, 1500, 2500, 5000])) if m.hostphase else int(rng.choice([300, 1000]))
                if not m.hostphase and m.unfetched + n > 1900:
                    c = int(rng.integers(0, len(contexts)))
                    log.append(("fetch", c, i))
                    contexts[c].fetch_ocean_state(m.p)
                    check_phase(m, "fetch before a burst")
                    m.unfetched = 0
                m.unfetched += n
                log.append(("burst", i, n))
                dt = np.float32(1 / 60)
                for _ in range(n):
                    m.p.update_ocean(float(dt))
                    oracle.update(m.phase, m.wavescale, dt)
                if m.hostphase:
                    check_phase(m, "hostphase after a burst")
            elif op == "displace":
                c = int(rng.integers(0, len(contexts)))
                log.append(("displace", c, i))
                contexts[c].displace_ocean_surface(m.p)
                got = contexts[c].read_displacement()
                ref = oracle.displace(np.asarray(m.p.height).copy(), m.phase.copy(), m.wavescale, m.choppiness, w=w)
                big = max(float(np.abs(ref[0]).max()), 1e-30)
                e = rmse(got[0][..., :3], ref[0][..., :3]) / big
                assert np.isfinite(got).all() and e < 2e-6, (q, "maps", c, i, e, log[-14:])
                worst = max(worst, e)
            elif op == "lerp":
                # (the steps issued so far used the old wave scale: the model applied them at once; from here on the new one)
                ws = float(np.exp(rng.uniform(np.log(6.0), np.log(300.0))))
                amp = float(0.0025 * 10.0 ** rng.uniform(-0.7, 0.7))
                t = float(rng.choice([1.0, 0.5, 0.1]))
                log.append(("lerp", i, ws, amp, t))
                m.p.lerp_ocean_waves(ws, amp, float(rng.uniform(4.0, 12.0)), (0.6, 0.8), t)
            elif op == "fetch":
                c = int(rng.integers(0, len(contexts)))
                log.append(("fetch", c, i))
                contexts[c].fetch_ocean_state(m.p)
                check_phase(m, "fetch")
                m.unfetched = 0
            elif op == "copy":
                if len(objs) >= 7:
                    continue
                log.append(("copy", i))
                objs.append(Model(m.p.copy(), m.phase.copy(), m.hostphase))
                objs[-1].unfetched = m.unfetched
            elif op == "reseed":
                log.append(("reseed", i))
                m.p.seed_ocean(7000 + 97 * q + k)
                m.phase = np.zeros((N, N), np.float32)
                m.unfetched = 0
                check_phase(m, "seed_ocean")
            elif op == "release":
                c = int(rng.integers(0, len(contexts)))
                keep = m.p if rng.random() < 0.5 else None
                log.append(("release", c, i if keep is not None else None))
                contexts[c].release_parked_states(keep)
            elif op == "new":
                if len(objs) >= 7:
                    continue
                log.append(("new",))
                objs.append(new_params(100 + k))

        # everything once more at the end: every params on every context, then its phase
        for c, ctx in enumerate(contexts):
            for i, m in enumerate(objs):
                log.append(("displace", c, i))
                ctx.displace_ocean_surface(m.p)
                got = ctx.read_displacement()
                ref = oracle.displace(np.asarray(m.p.height).copy(), m.phase.copy(), m.wavescale, m.choppiness, w=w)
                e = rmse(got[0][..., :3], ref[0][..., :3]) / max(float(np.abs(ref[0]).max()), 1e-30)
                assert e < 2e-6, (q, "maps at the end", c, i, e, log[-14:])
                ctx.fetch_ocean_state(m.p)
                check_phase(m, "fetch at the end")
    finally:
        for ctx in contexts:
            ctx.close()
    print(f"sequence {q:3d}: N={N:4d}, {len(contexts)} context(s), {len(objs)} OceanParams, {len(log)} calls: ok", flush=True)

print(f"{sequences} sequences ok; calls by kind {dict(sorted(counts.items()))}; worst displacement rmse / max {worst:.2e}")

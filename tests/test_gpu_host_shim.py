"""GPU tests of the C++ host shim (datum_amd/host/ocean.h) driven the way datum's example-ocean drives the reference
(examples/ocean/ocean.cpp): through the compiled example program and through the flat C view used by Python."""

import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = np.float32(1.0 / 60.0)


def rmse(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).mean()))


def _oracle_run(oracle, N, seed, frames):
    p = oracle.EXAMPLE
    _, h0 = oracle.seed(N, seed, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"], sanitize=True)
    phase = np.zeros((N, N), np.float32)
    sp, fl = 0.0, (0.0, 0.0)
    for _ in range(frames):
        oracle.update(phase, p["wavescale"], DT)
        sp, fl = oracle.update_scalars(p["swellspeed"], p["swelllength"], p["windspeed"], p["winddirection"], DT, sp, fl)
    maps = oracle.displace(h0, phase.copy(), p["wavescale"], p["choppiness"], w=oracle.weights(N, reduced=True))
    s = oracle.example_oceanset(N, swellphase=sp)
    return phase, sp, fl, maps, s


@pytest.mark.parametrize("N,frames", [(64, 60), (256, 5)])
def test_example_program(oracle, N, frames):
    exe = os.path.join(ROOT, "examples", "ocean_headless")
    assert os.path.exists(exe), "build it with `make examples`"
    out = subprocess.run([exe, str(N), str(frames), "1000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    vals = {}
    for line in out.stdout.splitlines():
        toks = line.split()
        key, nums = None, []
        for tok in toks:
            try:
                nums.append(float(tok))
            except ValueError:
                if key is not None and nums:
                    vals[key] = nums
                key, nums = tok, []
        if key is not None and nums:
            vals[key] = nums
    phase, sp, fl, maps, s = _oracle_run(oracle, N, 1000, frames)
    assert vals["swellphase"][0] == pytest.approx(sp, abs=1e-7)
    assert vals["flow"] == pytest.approx(list(fl), rel=1e-6)
    assert np.float32(vals["phase[10][20]"][0]) == phase[10, 20]  # bit-exact device phase, printed with 9 digits
    assert np.float32(vals[f"phase[{N-1}][{N-1}]"][0]) == phase[N - 1, N - 1]
    assert vals["dz_rms"][0] == pytest.approx(float(np.sqrt((maps[0, ..., 2].astype(np.float64) ** 2).mean())), rel=1e-5)
    assert vals["map[0][5][7]"] == pytest.approx(list(maps[0, 5, 7, :3]), abs=5e-6)
    assert vals["map[1][5][7]"] == pytest.approx(list(maps[1, 5, 7, :3]), abs=5e-6)
    v = oracle.gen(s, maps, 64, 64)[40, 33]
    got = vals["pos"] + vals["uv"] + vals["n"] + vals["t"]
    assert np.allclose(got, v, rtol=2e-4, atol=2e-4)


def test_farm_example_program(oracle):
    # examples/ocean_farm: the tile farm from C++ alone -- the parent forks its ranks before any HIP call and relays rank 0's
    # communicator id over pipes, every rank steps its tile and the field is reassembled through datum_ocean_farm_* (RCCL inside
    # the C ABI; no Python, no torch.distributed in those processes).  One rank here (one GPU per box); the payload it prints a
    # checksum of is the oracle's displacement field after the same steps.
    exe = os.path.join(ROOT, "examples", "ocean_farm")
    assert os.path.exists(exe), "build it with `make examples`"
    N, batches, steps = 256, 3, 4
    out = subprocess.run([exe, "1", str(N), str(batches), str(steps), "1"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("rank 0 batch")]
    assert len(lines) == batches and "own-tile ok" in lines[-1] and out.stdout.rstrip().endswith(": ok")
    sums = [l.split("tiles")[1].split()[0] for l in lines]
    assert len(set(sums)) == batches                          # every batch gathered a different field

    # the first batch against the HIP module driven from here (same seed, same steps): FNV-1a of the xyz32 payload
    from datum_amd import capi, host_api

    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    with capi.Ocean(N, 1) as oc:
        oc.set_cascade(0, 22.0, 1.35)
        oc.upload_state(0, p.height)
        for _ in range(steps):
            oc.update(np.float32(1 / 60))
            oc.displace()
        want = oc.read_maps(0)[0, ..., :3].astype(np.float32).tobytes()
    h = 1469598103934665603
    for b in want:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert sums[0] == f"{h:016x}"


def test_farm_example_two_ranks_on_one_gpu_fail_cleanly():
    # Two ranks of examples/ocean_farm sent to ONE device (DATUM_FARM_DEVICES=1): the id travels from rank 0 through the parent to
    # rank 1, both processes meet in RCCL's bootstrap -- and RCCL refuses two ranks on one GPU.  What must come out of that:
    # datum_ocean_farm_init returns DATUM_OCEAN_ECOMM (-6) on both ranks with RCCL's own text in datum_ocean_last_error, nothing
    # hangs, the program exits 1.  (The only part of a multi-rank farm a one-GPU box can run.)
    exe = os.path.join(ROOT, "examples", "ocean_farm")
    assert os.path.exists(exe), "build it with `make examples`"
    out = subprocess.run([exe, "2", "256", "1", "2"], capture_output=True, text=True, timeout=180,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DATUM_FARM_DEVICES="1"))
    assert out.returncode == 1
    assert "FAILED" in out.stdout
    for rank in (0, 1):
        line = [l for l in out.stderr.splitlines() if l.startswith(f"rank {rank}: datum_ocean_farm_init")]
        assert line and "(-6)" in line[0] and "ncclCommInitRank" in line[0] and "Duplicate GPU" in line[0], out.stderr


def test_render_through_host_api_and_wave_change(oracle):
    # seed, tick, render; then change the wind (lerp_ocean_waves recomputes h0, ocean.cpp:185-213) and keep going:
    # the device keeps its phase, takes the new h0, and still matches the oracle run the same way
    from datum_amd import host_api

    N = 128
    e = oracle.EXAMPLE
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1003)
    phase = np.zeros((N, N), np.float32)
    with host_api.OceanContext(N) as ctx:
        mesh = ctx.create_ocean(32, 48)
        for _ in range(7):
            p.update_ocean(DT)
            oracle.update(phase, e["wavescale"], DT)
        ctx.render_ocean_surface(mesh, p)
        m1 = ctx.read_displacement()
        want1 = oracle.displace(p.height.copy(), phase.copy(), e["wavescale"], e["choppiness"], w=oracle.weights(N, reduced=True))
        assert rmse(m1[..., :3], want1[..., :3]) < 1e-5

        p.lerp_ocean_waves(30.0, 0.004, 12.0, (0.6, 0.8), 0.5)
        s = p.scalars()
        for _ in range(3):
            p.update_ocean(DT)
            oracle.update(phase, s.wavescale, DT)
        ctx.render_ocean_surface(mesh, p)
        ctx.fetch_ocean_state(p)
        assert np.array_equal(p.phase, phase)
        m2 = ctx.read_displacement()
        want2 = oracle.displace(p.height.copy(), phase.copy(), s.wavescale, s.choppiness, w=oracle.weights(N, reduced=True))
        assert rmse(m2[..., :3], want2[..., :3]) < 1e-5
        assert rmse(m2[..., :3], m1[..., :3]) > 1e-3  # the sea really changed

        verts = ctx.read_vertices(mesh)
        so = oracle.OceanSet.from_buffer_copy(bytes(p.oceanset()))
        vwant = oracle.gen(so, m2, 32, 48)
        assert (np.abs(verts - vwant) / (1 + np.abs(vwant))).max() < 2e-4
        idx = ctx.read_indices(mesh)
        assert np.array_equal(idx, oracle.indices(32, 48))


def test_the_reference_configuration_in_literal_mode_against_the_golden_maps(oracle):
    # The reference as shipped: WaveResolution 64, example-ocean parameters, seed 1000, dt = 1/60 -- through the C++ mirror of datum's API with
    # OceanContext::literaltransform (the reference's own radix-2 transforms and literal twiddle table on the GPU): the golden maps of
    # tests/golden/ocean_n64.npz (the oracle with that same table) after 60 and after 600 ticks, to fp32 rounding
    from datum_amd import host_api

    g = np.load(os.path.join(ROOT, "tests", "golden", "ocean_n64.npz"))
    N = 64
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    assert np.array_equal(p.height, g["h0"])
    with host_api.OceanContext(N, literaltransform=True) as lit, host_api.OceanContext(N) as fused:
        mesh, mesh2 = lit.create_ocean(32, 32), fused.create_ocean(32, 32)
        done = 0
        for ticks in (60, 600):
            while done < ticks:
                p.update_ocean(DT)
                done += 1
            lit.render_ocean_surface(mesh, p)
            fused.render_ocean_surface(mesh2, p)
            m, mf = lit.read_displacement(), fused.read_displacement()
            want = g[f"maps_{ticks}"]
            e_lit, e_fused = rmse(m[..., :3], want[..., :3]), rmse(mf[..., :3], want[..., :3])
            assert e_lit < 1e-7 and e_lit < e_fused < 1e-5, (ticks, e_lit, e_fused)
            assert float(np.abs(m[..., :3].astype(np.float64) - want[..., :3]).max()) < 5e-7
        lit.fetch_ocean_state(p)
        assert np.array_equal(p.phase, g["phase_600"])


@pytest.mark.parametrize("heightfp16", [False, True])
def test_the_fp16_formats_through_the_host_api(oracle, heightfp16):
    # OceanContext::spectrumfp16 (+ ::heightfp16: the module reads OceanParams::height as halves too, DATUM_OCEAN_SPECTRUM_FP16_H0) through
    # the C++ mirror: displacement within the stated fp16 tolerance of the fp32 context's (RMSE < 2e-3 of the largest |displacement|),
    # before and after lerp_ocean_waves changed every h0 (the module's half copy has to follow the new h0), the phase state bit for bit the
    # fp32 context's and the caller's h0 untouched
    from datum_amd import host_api

    N = 256
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1003)
    with host_api.OceanContext(N, spectrumfp16=True, heightfp16=heightfp16) as half, host_api.OceanContext(N) as full:
        for leg in range(2):
            for _ in range(3):
                p.update_ocean(DT)
            before = p.height.copy()
            half.displace_ocean_surface(p)
            full.displace_ocean_surface(p)
            m, mf = half.read_displacement(), full.read_displacement()
            big = float(np.abs(mf[..., :3]).max())
            e = rmse(m[..., :3], mf[..., :3])
            assert big > 0.05 and 1e-7 * big < e < 2e-3 * big, (leg, e / big)
            assert np.array_equal(p.height, before)
            if leg == 0:
                p.lerp_ocean_waves(40.0, 0.004, 11.0, (0.6, 0.8), 1.0)
                assert not np.array_equal(p.height, before)
        half.fetch_ocean_state(p)
        ph = p.phase.copy()
        full.fetch_ocean_state(p)
        assert np.array_equal(ph, p.phase) and float(np.abs(ph).max()) > 0


def test_device_side_spectrum_rebuild(oracle):
    # lerp_ocean_waves' h0 rebuild on the device from the resident seed.
    # Tolerance: expf / division differ from libm by ulps -> 2e-6 relative to the largest |h0| (and exact zeros
    # where phillips is 0); the maps that follow stay within the 1e-5 RMSE bar.
    from datum_amd import capi, host_api

    N = 256
    e = oracle.EXAMPLE
    seed, h0 = oracle.seed(N, 1001, e["wavescale"], e["waveamplitude"], e["windspeed"], e["winddirection"])
    with capi.Ocean(N, 1) as oc:
        oc.set_cascade(0, e["wavescale"], e["choppiness"])
        oc.upload_seed(0, seed)
        oc.rebuild_height(0, e["wavescale"], e["waveamplitude"], e["windspeed"], e["winddirection"])
        got = oc.read_height(0)
        assert np.abs(got - h0).max() <= 2e-6 * np.abs(h0).max()
        assert np.all(got[N // 2, N // 2] == 0)
        want2 = oracle.height_from_seed(seed, 30.0, 0.004, 12.0, (0.6, 0.8))
        oc.rebuild_height(0, 30.0, 0.004, 12.0, (0.6, 0.8))
        got2 = oc.read_height(0)
        assert np.abs(got2 - want2).max() <= 2e-6 * np.abs(want2).max()
        oc.update(DT)
        oc.displace()
        phase = np.zeros((N, N), np.float32)
        oracle.update(phase, 30.0, DT)
        assert np.array_equal(oc.read_state(0), phase)  # the new wave scale drives the dispersion too
        ref = oracle.displace(want2, phase.copy(), 30.0, e["choppiness"], w=oracle.weights(N, reduced=True))
        assert rmse(oc.read_maps(0)[..., :3], ref[..., :3]) < 1e-5

    # the same through the C++ shim: deviceheight makes lerp_ocean_waves skip the host loop
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.set_deviceheight(True)
    p.seed_ocean(1001)
    with host_api.OceanContext(N) as ctx:
        p.update_ocean(DT)
        ctx.displace_ocean_surface(p)
        before = p.height.copy()
        p.lerp_ocean_waves(30.0, 0.004, 12.0, (0.6, 0.8), 1.0)
        assert np.array_equal(p.height, before)  # host copy untouched
        p.update_ocean(DT)
        ctx.displace_ocean_surface(p)
        ctx.fetch_ocean_state(p)
        s = p.scalars()
        want = oracle.height_from_seed(p.seed.copy(), s.wavescale, s.waveamplitude, s.windspeed, tuple(s.winddirection))
        assert np.abs(p.height - want).max() <= 2e-6 * np.abs(want).max()
        ph = np.zeros((N, N), np.float32)
        oracle.update(ph, e["wavescale"], DT)
        oracle.update(ph, s.wavescale, DT)
        assert np.array_equal(p.phase, ph)


def test_render_from_per_frame_copies(oracle):
    # The reference's OceanParams is a POD that game code may copy (e.g. into a double-buffered render state every frame).
    # The update history belongs to the state, not to one object: rendering from a fresh copy each frame applies every
    # update_ocean step exactly once (not k * dt on frame k, and not zero), a never-seeded OceanParams renders a flat
    # ocean, and a step issued before lerp_ocean_waves is advanced with the wave scale in force at that step.
    import numpy as np

    from datum_amd import host_api

    N = 128
    dt = np.float32(1 / 60)
    e = oracle.EXAMPLE
    params = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    params.seed_ocean(1000)
    phase = np.zeros((N, N), np.float32)
    with host_api.OceanContext(N, device=0) as ctx:
        for frame in range(6):
            params.update_ocean(dt)
            oracle.update(phase, e["wavescale"], dt)
            snapshot = params.copy()                  # what a render thread would be handed
            ctx.displace_ocean_surface(snapshot)
            del snapshot
        # a step under the old wave scale, then the wind changes, then another step: each under its own dispersion
        params.update_ocean(dt)
        oracle.update(phase, e["wavescale"], dt)
        params.lerp_ocean_waves(64.0, e["waveamplitude"], e["windspeed"], e["winddirection"], 1.0)
        ws = float(params.scalars().wavescale)
        assert ws == 64.0
        params.update_ocean(dt)
        oracle.update(phase, ws, dt)
        ctx.displace_ocean_surface(params.copy())
        ctx.fetch_ocean_state(params)
        assert np.array_equal(params.phase, phase)
        assert params.scalars().pending == 0          # the host phase now contains the history

    # never seeded: zero state, flat ocean, no error
    blank = host_api.OceanParams(64)
    with host_api.OceanContext(64, device=0) as ctx:
        blank.update_ocean(dt)
        ctx.displace_ocean_surface(blank)
        m = ctx.read_displacement()
        assert np.all(m[0] == 0) and np.abs(m[1][..., 2] - 1).max() < 1e-6


def test_two_oceans_on_one_context_for_longer_than_the_history(oracle):
    # Two OceanParams rendered alternately through ONE OceanContext, as two oceans (or several cascades driven through the
    # host API) would be, for more update_ocean calls than the history records (OceanParams::MaxRecordedUpdates = 4096).
    # The reference keeps each params' phase on the host, so any context renders any params at any time
    # (ocean.cpp:217-236); here the context parks the state it is not rendering on the device and continues from there:
    # every step applied exactly once to the right state, no growth of the work per frame, no exception.
    import time

    import numpy as np

    from datum_amd import host_api

    N = 64
    dt = np.float32(1 / 60)
    ws = (22.0, 64.0)
    ps = []
    for k in range(2):
        p = host_api.OceanParams(N, **dict(host_api.EXAMPLE_TUNABLES, wavescale=ws[k]))
        p.set_hostphase(False)             # (on by default at 64 x 64: this test is about the phase that lives on the device only)
        p.seed_ocean(1000 + k)
        ps.append(p)
    phases = [np.zeros((N, N), np.float32) for _ in range(2)]
    steps = 4400
    laps = []
    with host_api.OceanContext(N, device=0) as ctx:
        for frame in range(steps):
            t0 = time.perf_counter()
            for k in range(2):
                ps[k].update_ocean(dt)
                ctx.displace_ocean_surface(ps[k])
            if frame in (99, steps - 1):
                ctx.read_displacement()            # drain the stream: the lap times are host times
            laps.append(time.perf_counter() - t0)
        for k in range(2):
            for _ in range(steps):
                oracle.update(phases[k], ws[k], dt)
            ctx.fetch_ocean_state(ps[k])
            assert np.array_equal(ps[k].phase, phases[k]), k
        # the last frames cost what the first frames cost (round 2: a replay of the whole history on every switch)
        early, late = np.median(laps[100:400]), np.median(laps[-300:])
        assert late < 3 * early + 1e-4, (early, late)
        # and the maps are the second ocean's after rendering the second ocean
        want = oracle.displace(ps[1].height.copy(), phases[1].copy(), ws[1], 1.35, w=oracle.weights(N))
        ctx.displace_ocean_surface(ps[1])
        got = ctx.read_displacement()
        assert np.sqrt(((got.astype(np.float64) - want) ** 2).mean()) < 1e-5


def test_six_oceans_round_robin_on_one_context(oracle):
    # round-3 advisor finding: a context parks up to MaxParkedStates (4) states besides the one that is bound.  With more
    # states than that rendered round-robin, the least recently used slot is exactly the state that comes next: parking
    # before looking the resume slot up evicted it, every switch fell back to a host upload + a replay of the whole history,
    # and past the trimmed history it threw.  Now the resume slot is found first and kept out of the eviction.  Six states
    # (one bound + four parked + one that always has to come from the host) for longer than the history records: the five
    # that fit keep their device copies (after the sixth intruded, one lap of host uploads + replays, then slots again -- a
    # state evicted AFTER its history was trimmed could not come back without hostphase, as ocean.h documents); the phase of
    # every one is bit-exact; release_parked_states gives the memory back.
    import numpy as np

    from datum_amd import host_api

    N = 64
    dt = np.float32(1 / 60)
    K = 6
    ws = [22.0 + 7.0 * k for k in range(K)]
    ps = []
    for k in range(K):
        p = host_api.OceanParams(N, **dict(host_api.EXAMPLE_TUNABLES, wavescale=ws[k]))
        p.seed_ocean(1000 + k)
        p.set_hostphase(k == K - 1)        # the one state without a slot lives on its host copy, as the reference's all do
        ps.append(p)
    phases = [np.zeros((N, N), np.float32) for _ in range(K)]
    steps = 4400                          # more update_ocean calls per state than OceanParams::MaxRecordedUpdates records
    with host_api.OceanContext(N, device=0) as ctx:
        for frame in range(steps):
            for k in range(5):
                ps[k].update_ocean(dt)
                ctx.displace_ocean_surface(ps[k])
            if frame in (0, 600):         # the sixth state intrudes while the others' histories still reach their host copies
                ps[5].update_ocean(dt)
                oracle.update(phases[5], ws[5], dt)
                ctx.displace_ocean_surface(ps[5])
        assert ctx.parked_states() == 4
        for k in range(5):
            for _ in range(steps):
                oracle.update(phases[k], ws[k], dt)
        for k in range(K):
            ctx.fetch_ocean_state(ps[k])
            assert np.array_equal(ps[k].phase, phases[k]), k
        freed = ctx.release_parked_states(keep=ps[0])
        assert freed > 0 and freed % (12 * N * N) == 0
        assert ctx.parked_states() <= 1
        ctx.release_parked_states()
        assert ctx.parked_states() == 0
        # and the states are still renderable (host copy as of the fetch + recorded history)
        ps[2].update_ocean(dt)
        oracle.update(phases[2], ws[2], dt)
        ctx.displace_ocean_surface(ps[2])
        ctx.fetch_ocean_state(ps[2])
        assert np.array_equal(ps[2].phase, phases[2])


@pytest.mark.parametrize("hostphase", [False, True])
def test_diverged_copies_share_an_id_but_not_a_history(oracle, hostphase):
    # copy P to Q, then advance them differently: the reference's PODs diverge freely.  Both carry the same state id and
    # history numbers; the context must notice (lineage of the last applied entry) and render each with its own phase.
    import numpy as np

    from datum_amd import host_api

    N = 64
    e = oracle.EXAMPLE
    a, b = np.float32(1 / 60), np.float32(1 / 24)
    P = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    P.set_hostphase(hostphase)
    P.seed_ocean(1000)
    pp, pq = np.zeros((N, N), np.float32), np.zeros((N, N), np.float32)
    with host_api.OceanContext(N, device=0) as ctx:
        for _ in range(3):
            P.update_ocean(a)
            oracle.update(pp, e["wavescale"], a)
            oracle.update(pq, e["wavescale"], a)
            ctx.displace_ocean_surface(P)
        Q = P.copy()
        for _ in range(5):
            P.update_ocean(a)
            oracle.update(pp, e["wavescale"], a)
            Q.update_ocean(b)
            oracle.update(pq, e["wavescale"], b)
            ctx.displace_ocean_surface(P)
            ctx.displace_ocean_surface(Q)
        ctx.fetch_ocean_state(Q)
        ctx.fetch_ocean_state(P)
        assert np.array_equal(P.phase, pp)
        assert np.array_equal(Q.phase, pq)
        assert not np.array_equal(pp, pq)


def test_off_screen_for_longer_than_the_history_at_the_reference_resolution(oracle):
    # At the reference's own resolution (64 x 64) OceanParams::hostphase is on by default: an ocean that is rendered, then ticked
    # off-screen for longer than the history records (4500 update_ocean calls, 75 s at 60 Hz), then rendered again -- by the context
    # that still holds its (now unreachable) device copy, and by one that never saw it -- comes back with the right phase, as in the
    # reference, where every update_ocean advances the host copy (ocean.cpp:223-233).  Without hostphase this is the documented
    # failure of test_a_context_that_never_saw_the_state.
    import numpy as np

    from datum_amd import host_api

    N = 64
    dt = np.float32(1 / 60)
    e = oracle.EXAMPLE
    p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    phase = np.zeros((N, N), np.float32)
    w = oracle.weights(N)
    with host_api.OceanContext(N, device=0) as ctx, host_api.OceanContext(N, device=0) as other:
        for _ in range(10):
            p.update_ocean(dt)
            oracle.update(phase, e["wavescale"], dt)
            ctx.displace_ocean_surface(p)
        for _ in range(4500):
            p.update_ocean(dt)
            oracle.update(phase, e["wavescale"], dt)
        assert np.array_equal(p.phase, phase)              # the host copy followed every call
        assert p.to_pod() is not None
        want = oracle.displace(p.height.copy(), phase.copy(), e["wavescale"], e["choppiness"], w=w)
        for c in (ctx, other):
            c.displace_ocean_surface(p)
            got = c.read_displacement()
            assert np.sqrt(((got.astype(np.float64) - want) ** 2).mean()) < 1e-5
            c.fetch_ocean_state(p)
            assert np.array_equal(p.phase, phase)


def test_a_context_that_never_saw_the_state(oracle):
    # 4500 update_ocean calls without a render or a fetch: the history no longer reaches back to params.phase.  With
    # OceanParams::hostphase the host copy is advanced as the reference does it and any context can start from it;
    # without, the first render by a context that holds no copy fails loudly instead of rendering a wrong phase.
    import numpy as np

    from datum_amd import host_api

    N = 64
    dt = np.float32(1 / 60)
    e = oracle.EXAMPLE
    steps = 4500
    kept = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    kept.seed_ocean(1000)
    kept.set_hostphase(True)
    lost = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
    lost.set_hostphase(False)                         # (on by default at 64 x 64, where nothing below could be lost)
    lost.seed_ocean(1000)
    phase = np.zeros((N, N), np.float32)
    for _ in range(steps):
        kept.update_ocean(dt)
        lost.update_ocean(dt)
        oracle.update(phase, e["wavescale"], dt)
    assert np.array_equal(kept.phase, phase)          # the host loop is the reference's, bit for bit the oracle's
    with host_api.OceanContext(N, device=0) as ctx:
        ctx.displace_ocean_surface(kept)
        ctx.fetch_ocean_state(kept)
        assert np.array_equal(kept.phase, phase)
        with pytest.raises(RuntimeError) as err:
            ctx.displace_ocean_surface(lost)
        assert "hostphase" in str(err.value)

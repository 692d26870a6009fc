"""CPU tests of the C++ host shim (datum_amd/host/ocean.h): the host-side functions of the path
(seed_ocean, lerp_ocean_*, update_ocean's scalar part, the OceanSet header, the twiddle table) against the
oracle's restatement of src/renderer/ocean.cpp:82-236,686-700,731-747.  Nothing here touches a GPU."""

import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def host():
    from datum_amd import host_api

    host_api.load()
    return host_api


@pytest.mark.parametrize("N,rngseed", [(64, 1000), (64, 1003), (256, 1001)])
def test_seed_ocean_matches_oracle(host, oracle, N, rngseed):
    p = host.OceanParams(N, **host.EXAMPLE_TUNABLES)
    p.seed_ocean(rngseed)
    s = p.scalars()
    e = oracle.EXAMPLE
    want_seed, want_h0, rej = oracle.seed(N, rngseed, e["wavescale"], e["waveamplitude"], e["windspeed"], e["winddirection"], sanitize=True, return_rejected=True)
    assert s.rejectedseeds == rej
    assert np.array_equal(p.seed, want_seed)
    assert np.array_equal(p.height, want_h0)
    assert np.all(p.phase == 0)
    assert s.swellphase == 0 and tuple(s.flow) == (0, 0)


def test_seed_ocean_random_device(host):
    a, b = host.OceanParams(64), host.OceanParams(64)
    a.seed_ocean()
    b.seed_ocean()
    assert np.isfinite(a.height).all()
    assert not np.array_equal(a.seed, b.seed)
    assert abs(float(a.seed.std()) - 1.0) < 0.1  # unit complex Gaussian per component


def test_defaults_are_the_reference_defaults(host):
    s = host.OceanParams(64).scalars()  # src/renderer/ocean.h:50-65
    assert tuple(s.plane) == (0, 0, 1, 0)
    assert (s.swelllength, s.swellsteepness, s.swellspeed) == (40.0, 0.0, 1.25)
    assert s.swellamplitude == np.float32(0.8)
    assert tuple(s.swelldirection) == (np.float32(0.780869), np.float32(0.624695))
    assert (s.wavescale, s.windspeed, s.smoothing) == (64.0, 30.0, 280.0)
    assert s.waveamplitude == np.float32(0.00002)
    assert s.choppiness == np.float32(1.35)
    assert s.resolution == 64


def test_lerp_ocean_waves_recomputes_h0(host, oracle):
    N = 64
    p = host.OceanParams(N, **host.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    before = p.height.copy()
    e = oracle.EXAMPLE
    # same targets: nothing happens (ocean.cpp:187)
    p.lerp_ocean_waves(e["wavescale"], np.float32(e["waveamplitude"]), np.float32(e["windspeed"]), tuple(np.float32(v) for v in e["winddirection"]), 0.5)
    assert np.array_equal(p.height, before)
    p.lerp_ocean_waves(30.0, 0.004, 12.0, (0.6, 0.8), 0.25)
    s = p.scalars()
    lerp = lambda a, b, t: oracle.lib().oracle_lerp(a, b, t)
    assert s.wavescale == np.float32(lerp(22.0, 30.0, 0.25))
    assert s.windspeed == np.float32(lerp(np.float32(7.9), 12.0, 0.25))
    wd = np.array([lerp(np.float32(0.780869), np.float32(0.6), 0.25), lerp(np.float32(0.624695), np.float32(0.8), 0.25)], np.float32)
    wd = wd / np.sqrt(wd[0] * wd[0] + wd[1] * wd[1], dtype=np.float32)
    assert np.allclose(tuple(s.winddirection), wd, rtol=1e-6)
    want = oracle.height_from_seed(p.seed.copy(), s.wavescale, s.waveamplitude, s.windspeed, tuple(s.winddirection))
    assert np.array_equal(p.height, want)
    assert not np.array_equal(p.height, before)


def test_lerp_ocean_swell(host):
    p = host.OceanParams(64)
    p.lerp_ocean_swell(40.0, np.float32(0.8), 1.25, (np.float32(0.780869), np.float32(0.624695)), 0.3)
    s = p.scalars()
    assert (s.swelllength, s.swellspeed) == (40.0, 1.25)
    p.lerp_ocean_swell(60.0, 1.0, 2.0, (1.0, 0.0), 0.5)
    s = p.scalars()
    assert s.swelllength == 50.0 and abs(s.swellamplitude - 0.9) < 1e-6 and s.swellspeed == 1.625
    d = np.array(s.swelldirection)
    assert abs(np.linalg.norm(d) - 1) < 1e-6


def test_update_ocean_scalars_and_queue(host, oracle):
    p = host.OceanParams(64, **host.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    e = oracle.EXAMPLE
    sp, fl = 0.0, (0.0, 0.0)
    for i in range(50):
        dt = np.float32(1 / 60 + 0.001 * i)
        p.update_ocean(dt)
        sp, fl = oracle.update_scalars(e["swellspeed"], e["swelllength"], e["windspeed"], e["winddirection"], dt, sp, fl)
    s = p.scalars()
    assert s.swellphase == np.float32(sp)
    assert tuple(s.flow) == tuple(np.float32(v) for v in fl)
    assert s.pending == 50  # the device's phase advance waits for the next render: 50 recorded steps
    # at the reference's own resolution the host copy of the phase follows every call, as in the reference (hostphase defaults to on
    # for N <= 64): bit for bit the oracle's loop
    phase = np.zeros((64, 64), np.float32)
    for i in range(50):
        oracle.update(phase, e["wavescale"], np.float32(1 / 60 + 0.001 * i))
    assert np.array_equal(p.phase, phase)
    # switched off (and by default above 64 x 64) the phase lives on the device only
    for q in (host.OceanParams(64, **host.EXAMPLE_TUNABLES), host.OceanParams(128, **host.EXAMPLE_TUNABLES)):
        if q.N == 64:
            q.set_hostphase(False)
        q.seed_ocean(1000)
        for i in range(5):
            q.update_ocean(np.float32(1 / 60))
        assert q.scalars().pending == 5 and np.all(q.phase == 0)


def test_oceanset_header_matches_oracle(host, oracle):
    N = 64
    p = host.OceanParams(N, **host.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    for _ in range(7):
        p.update_ocean(np.float32(1 / 60))
    got = p.oceanset()
    want = oracle.example_oceanset(N, swellphase=p.scalars().swellphase)
    for name, _ in got._fields_:
        a = np.array(getattr(got, name))
        b = np.array(getattr(want, name))
        assert np.allclose(a, b, rtol=2e-7, atol=1e-7), name
    # camera at (0,0,8): translation comes back out of the dual quaternion (transform.h:39)
    r = np.array(got.camera_real)
    d = np.array(got.camera_dual)
    conj = r * [1, -1, -1, -1]

    def qmul(a, b):
        return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                         a[0]*b[2]+a[2]*b[0]+a[3]*b[1]-a[1]*b[3], a[0]*b[3]+a[3]*b[0]+a[1]*b[2]-a[2]*b[1]])

    assert np.allclose(2 * qmul(d, conj)[1:], (0, 0, 8), atol=1e-5)
    # proj * invproj == identity
    assert np.allclose(np.array(got.proj).reshape(4, 4) @ np.array(got.invproj).reshape(4, 4), np.eye(4), atol=1e-5)
    assert ctypes.sizeof(got) == 216


def test_twiddle_table_is_the_reference_formula(host, oracle):
    for N in (64, 512):
        assert np.array_equal(host.twiddle_table(N), oracle.weights(N))


def test_hostphase_is_the_reference_loop(host, oracle):
    # OceanParams::hostphase: update_ocean also advances the host copy of the phase with the reference's own loop
    # (ocean.cpp:223-233), bit for bit the oracle's -- through a change of wave scale and past the point where the recorded
    # history (OceanParams::MaxRecordedUpdates = 4096 entries) is trimmed.  Without it the host copy stays where it was.
    N = 32
    dt = np.float32(1 / 60)
    e = oracle.EXAMPLE
    p = host.OceanParams(N, **host.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    p.set_hostphase(True)
    q = p.copy()
    q.set_hostphase(False)
    phase = np.zeros((N, N), np.float32)
    for i in range(4300):
        if i == 100:
            p.lerp_ocean_waves(64.0, e["waveamplitude"], e["windspeed"], e["winddirection"], 1.0)
        ws = float(p.scalars().wavescale)
        p.update_ocean(dt)
        q.update_ocean(dt)
        oracle.update(phase, ws, dt)
    assert np.array_equal(p.phase, phase)
    assert np.all(q.phase == 0)
    assert p.scalars().pending <= 4096 and q.scalars().pending <= 4096


def test_hostphase_switched_on_after_the_history_was_trimmed(host):
    # round-3 advisor finding: hostphase is a public bool; switched on after more than MaxRecordedUpdates steps without it
    # (and without a fetch) the steps between the host phase and the recorded history are gone.  update_ocean used to index
    # updates[phaseupdates - firstupdate] with phaseupdates < firstupdate (an unsigned underflow: out of bounds).  Now the
    # setter refuses, and the bare field -- as C++ code would set it -- makes update_ocean throw and leave the params as it was.
    p = host.OceanParams(32, **host.EXAMPLE_TUNABLES)
    p.set_hostphase(False)                        # (on by default at N <= 64)
    p.seed_ocean(1000)
    dt = np.float32(1 / 60)
    for _ in range(4300):
        p.update_ocean(dt)
    pending = p.scalars().pending
    assert pending < 4300                         # the history was trimmed
    with pytest.raises(host.HostError, match="fetch_ocean_state"):
        p.set_hostphase(True)
    p.poke_hostphase(True)
    sp = p.scalars().swellphase
    with pytest.raises(host.HostError, match="hostphase was set after"):
        p.update_ocean(dt)
    assert p.scalars().pending == pending         # the refused step is not recorded, nothing was advanced
    assert p.scalars().swellphase == sp
    assert np.all(p.phase == 0)
    p.poke_hostphase(False)
    p.update_ocean(dt)                            # and the params goes on as before
    assert p.scalars().pending == pending + 1
    assert p.scalars().swellphase != sp
    # switched on in time (history intact) it catches up in one call
    q = host.OceanParams(32, **host.EXAMPLE_TUNABLES)
    q.set_hostphase(False)
    q.seed_ocean(1000)
    for _ in range(100):
        q.update_ocean(dt)
    q.set_hostphase(True)
    q.update_ocean(dt)
    assert np.any(q.phase != 0)


def test_reference_pod_round_trip(host):
    # OceanParamsPod: the reference's OceanParams byte for byte (src/renderer/ocean.h:48-73 with WaveResolution = 64) for callers
    # that memcpy / serialise a params by value.  Field offsets as the reference's struct lays them out; a round trip gives a
    # fresh state with the same tunables, scalars and arrays; with steps recorded that the host phase does not hold, to_pod says so.
    p = host.OceanParams(64, **host.EXAMPLE_TUNABLES)
    p.seed_ocean(1000)
    pod = p.to_pod()
    assert pod is not None and len(pod) == 82000
    f = np.frombuffer(pod, np.float32)
    # plane (0,0,1,0) | swell length, amplitude, steepness, speed, direction | waves scale, amplitude, windspeed, direction, choppiness, smoothing | swellphase
    assert tuple(f[:4]) == (0, 0, 1, 0)
    assert f[4] == 40.0 and f[5] == np.float32(0.8) and f[7] == 1.25
    assert f[10] == 22.0 and f[11] == np.float32(0.0025) and f[12] == np.float32(7.9)
    assert f[15] == np.float32(1.35) and f[16] == 320.0 and f[17] == 0.0
    o = 18
    assert np.array_equal(f[o:o + 8192].reshape(64, 64, 2), p.seed)
    assert np.array_equal(f[o + 8192:o + 16384].reshape(64, 64, 2), p.height)
    assert np.all(f[o + 16384:o + 20480] == 0)                       # phase
    assert tuple(f[o + 20480:]) == (0, 0)                             # flow
    q = host.OceanParams.from_pod(pod)
    assert np.array_equal(q.seed, p.seed) and np.array_equal(q.height, p.height) and np.array_equal(q.phase, p.phase)
    assert bytes(q.scalars())[:ctypes.sizeof(host.Scalars) - 12] == bytes(p.scalars())[:ctypes.sizeof(host.Scalars) - 12]
    assert q.to_pod() == pod
    p.update_ocean(np.float32(1 / 60))                                # the host copy is current after every call (hostphase: the default at 64 x 64)
    pod2 = p.to_pod()
    assert pod2 is not None and np.any(np.frombuffer(pod2, np.float32)[o + 16384:o + 20480] != 0)
    h = host.OceanParams(64, **host.EXAMPLE_TUNABLES)
    h.set_hostphase(False)
    h.seed_ocean(1000)
    assert h.to_pod() is not None
    h.update_ocean(np.float32(1 / 60))                                # without it: recorded, not in the host phase -- the POD would be stale
    assert h.to_pod() is None
    with pytest.raises(host.HostError):
        host.OceanParams(128).to_pod()

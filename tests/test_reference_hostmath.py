"""The oracle and the C++ host shim against the REFERENCE'S OWN host-math lines (CPU).

tests/golden/ref_hostmath_n64.npz is data produced by tests/golden/make_ref_hostmath.py in the build container: the text of
/root/reference/src/renderer/ocean.cpp:80-236 (dispersion, phillips, guass_random_distribution, seed_ocean, lerp_ocean_swell,
lerp_ocean_waves, update_ocean) and the OceanParams struct of ocean.h:48-73, read at run time, compiled between stand-ins for leap's
lml (datum_amd/host/lml.h), for OceanContext and for random_device (a fixed seed), once as the reference's build does
(-ffast-math, src/CMakeLists.txt:10) and once without.  Not a reference build in the rubric's sense (stand-in headers: DESIGN.md
section 3) -- but what the oracle and the shim are compared with here is what the reference's lines computed, not what the oracle
says they compute: SURVEY.md 8(a) rows a2-a6, 7 step 1 (iii)."""

import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DT = np.float32(1.0 / 60.0)


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(HERE, "golden", "ref_hostmath_n64.npz"))


@pytest.fixture(scope="module")
def host():
    from datum_amd import host_api

    host_api.load()
    return host_api


def _fields(ref, key):
    return dict(zip([str(n) for n in ref["dump_fields"]], ref[key]))


def test_fixture_is_the_example_ocean(ref):
    # examples/ocean/ocean.cpp:46-50 over the defaults of ocean.h:50-65; no rejected Gaussian pair at this seed (DESIGN F7)
    f = _fields(ref, "params_seeded")
    assert (f["wavescale"], f["windspeed"]) == (22.0, np.float32(7.9)) and f["waveamplitude"] == np.float32(0.0025)
    assert (f["windx"], f["windy"]) == (np.float32(0.780869), np.float32(0.624695))
    assert f["swellphase"] == 0 and (f["flowx"], f["flowy"]) == (0, 0)
    assert ref["seed"].shape == (64, 64, 2) and np.isfinite(ref["seed"]).all()
    assert abs(float(ref["seed"].std()) - 1.0) < 0.05


def test_oracle_seed_and_h0_are_the_reference_lines(oracle, ref):
    # a4 + a5: mt19937 consumption order, the polar draw, phillips, dk * sqrt(P / 2) -- bit for bit the non-fast-math build
    e = oracle.EXAMPLE
    seed, h0, rej = oracle.seed(64, 1000, e["wavescale"], e["waveamplitude"], e["windspeed"], e["winddirection"], sanitize=False, return_rejected=True)
    assert rej == 0
    assert np.array_equal(seed, ref["seed"])
    assert np.array_equal(h0, ref["h0"])
    # the committed oracle-made fixture of the GPU tests starts from the same numbers
    g = np.load(os.path.join(HERE, "golden", "ocean_n64.npz"))
    assert np.array_equal(g["seed"], ref["seed"]) and np.array_equal(g["h0"], ref["h0"])
    # the reference's own build (-ffast-math): the seed within an ulp or two of log / sqrt, h0 within 1e-6 relative of its largest value
    assert np.abs(ref["fast_seed"] - ref["seed"]).max() <= 4 * np.spacing(np.float32(np.abs(ref["seed"]).max()))
    assert np.abs(ref["fast_h0"] - ref["h0"]).max() <= 1e-6 * np.abs(ref["h0"]).max()


def test_oracle_update_is_the_reference_loop(oracle, ref):
    # a3 + a6: fmod(phase + dispersion(k) dt, 2 pi) accumulated step by step, swellphase and flow beside it
    e = oracle.EXAMPLE
    phase = np.zeros((64, 64), np.float32)
    swellphase, flow = np.float32(0), (np.float32(0), np.float32(0))
    f0 = _fields(ref, "params_seeded")
    done = 0
    for steps in (1, 60, 600):
        for _ in range(steps - done):
            oracle.update(phase, e["wavescale"], DT)
            swellphase, flow = oracle.update_scalars(f0["swellspeed"], f0["swelllength"], f0["windspeed"], (f0["windx"], f0["windy"]), DT, swellphase, flow)
        done = steps
        assert np.array_equal(phase, ref[f"phase_{steps}"]), steps
        f = _fields(ref, f"params_{steps}")
        assert np.float32(swellphase) == f["swellphase"] and (np.float32(flow[0]), np.float32(flow[1])) == (f["flowx"], f["flowy"]), steps
    # the GPU tests' phase fixtures (made by the oracle) are therefore the reference's own numbers
    g = np.load(os.path.join(HERE, "golden", "ocean_n64.npz"))
    for steps in (1, 60, 600):
        assert np.array_equal(g[f"phase_{steps}"], ref[f"phase_{steps}"])
    # -ffast-math build: the accumulated phase stays within a few ulps of 2 pi per step (reassociated k, reciprocal of the wave scale)
    for steps, tol in ((1, 2e-6), (60, 4e-5), (600, 4e-4)):
        d = np.abs(ref[f"fast_phase_{steps}"].astype(np.float64) - ref[f"phase_{steps}"])
        d = np.minimum(d, 2 * np.pi - d)            # (a phase that wrapped one step earlier or later)
        assert d.max() < tol, (steps, d.max())


def test_oracle_lerp_waves_is_the_reference_rebuild(oracle, ref):
    # a5': the blend of the wave parameters and the h0 rebuild from the stored seed (ocean.cpp:185-213)
    call = ref["lerp_waves_call"]
    f0 = _fields(ref, "params_600")
    lerp = lambda a, b, t: np.float32(oracle.lib().oracle_lerp(a, b, t))
    ws = lerp(f0["wavescale"], call[0], call[5])
    wa = lerp(f0["waveamplitude"], call[1], call[5])
    wv = lerp(f0["windspeed"], call[2], call[5])
    f = _fields(ref, "params_lerped")
    assert (ws, wa, wv) == (f["wavescale"], f["waveamplitude"], f["windspeed"])
    h0 = oracle.height_from_seed(ref["seed"], f["wavescale"], f["waveamplitude"], f["windspeed"], (f["windx"], f["windy"]))
    assert np.array_equal(h0, ref["h0_lerped"])
    # ... and the update loop on the blended wave scale
    phase = ref["phase_600"].copy()
    for _ in range(2):
        oracle.update(phase, f["wavescale"], DT)
    assert np.array_equal(phase, ref["phase_602_lerped"])


def test_host_shim_is_the_reference_lines(host, ref):
    # the product's host side (datum_amd/host/ocean.cpp) through the same sequence of calls: seed_ocean(1000), 600 update_ocean,
    # lerp_ocean_waves, lerp_ocean_swell, 2 update_ocean -- every array and every scalar bit for bit the non-fast-math build
    p = host.OceanParams(64, **host.EXAMPLE_TUNABLES)
    p.set_hostphase(True)                                   # the host copy of the phase follows every call, as in the reference
    p.seed_ocean(1000)
    assert np.array_equal(p.seed, ref["seed"]) and np.array_equal(p.height, ref["h0"])
    assert p.scalars().rejectedseeds == 0

    def scalars_are(key):
        s, f = p.scalars(), _fields(ref, key)
        got = dict(wavescale=s.wavescale, waveamplitude=s.waveamplitude, windspeed=s.windspeed, windx=s.winddirection[0], windy=s.winddirection[1],
                   swelllength=s.swelllength, swellamplitude=s.swellamplitude, swellspeed=s.swellspeed, swelldirx=s.swelldirection[0], swelldiry=s.swelldirection[1],
                   swellphase=s.swellphase, flowx=s.flow[0], flowy=s.flow[1])
        for k, v in f.items():
            assert np.float32(got[k]) == v, (key, k, got[k], v)

    scalars_are("params_seeded")
    done = 0
    for steps in (1, 60, 600):
        for _ in range(steps - done):
            p.update_ocean(DT)
        done = steps
        assert np.array_equal(p.phase, ref[f"phase_{steps}"]), steps
        scalars_are(f"params_{steps}")
    c = [float(v) for v in ref["lerp_waves_call"]]
    p.lerp_ocean_waves(c[0], c[1], c[2], (c[3], c[4]), c[5])
    scalars_are("params_lerped")
    assert np.array_equal(p.height, ref["h0_lerped"])
    c = [float(v) for v in ref["lerp_swell_call"]]
    p.lerp_ocean_swell(c[0], c[1], c[2], (c[3], c[4]), c[5])
    scalars_are("params_swell")
    for _ in range(2):
        p.update_ocean(DT)
    assert np.array_equal(p.phase, ref["phase_602_lerped"])
    scalars_are("params_602_lerped")


def test_generator_reads_the_reference_and_keeps_none_of_it():
    # the script that made the fixture holds no reference text: it reads ocean.cpp:80-236 / ocean.h:48-73 at run time (rule: a fixture is data)
    text = open(os.path.join(HERE, "golden", "make_ref_hostmath.py")).read()
    assert 'lines(os.path.join(REF, "ocean.cpp"), 80, 236)' in text and 'lines(os.path.join(REF, "ocean.h"), 48, 73)' in text
    for needle in ("klength2", "kdotw", "real11(entropy)", "damping"):
        assert needle not in text, needle

"""CPU tests: the oracle reproduces the committed golden fixtures, and the C-ABI library loads and exports
every symbol include/datum_ocean_hip.h declares (no compute calls: there is no GPU here)."""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ocean_n64.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def test_oracle_reproduces_golden(oracle, golden):
    p = oracle.EXAMPLE
    N = 64
    seed, h0 = oracle.seed(N, 1000, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"], sanitize=False)
    assert np.array_equal(seed, golden["seed"])
    assert np.array_equal(h0, golden["h0"])
    phase = np.zeros((N, N), np.float32)
    done = 0
    for steps in (1, 60, 600):
        for _ in range(steps - done):
            oracle.update(phase, p["wavescale"], np.float32(1 / 60))
        done = steps
        assert np.array_equal(phase, golden[f"phase_{steps}"])
        m = oracle.displace(h0, phase.copy(), p["wavescale"], p["choppiness"], dt=0.0)
        assert np.array_equal(m, golden[f"maps_{steps}"])
    s = oracle.example_oceanset(N, swellphase=0.25)
    assert bytes(s) == golden["oceanset"].tobytes()
    v = oracle.gen(s, golden["maps_600"], 32, 32)
    assert np.allclose(v, golden["vertices_600_32x32"], rtol=0, atol=1e-6)
    import gen_cases

    steep = gen_cases.oceanset(oracle, N, "pitched_steep", swellphase=1.9)
    assert steep.swellsteepness > 0.5
    assert bytes(steep) == golden["oceanset_steep"].tobytes()
    v = oracle.gen(steep, golden["maps_600"], 48, 40)
    assert np.allclose(v, golden["vertices_600_steep_48x40"], rtol=0, atol=1e-6)
    for name in ("rolled", "plane_w"):
        hdr = gen_cases.oceanset(oracle, N, name, swellphase=0.9)
        assert bytes(hdr) == golden[f"oceanset_{name}"].tobytes()
        v = oracle.gen(hdr, golden["maps_600"], 40, 30)
        # positions reach 1e6 where the rays miss the plane: compare as the stated tolerance does
        pos, tex, frame = gen_cases.compare(v, golden[f"vertices_600_{name}_40x30"])
        assert pos < 1e-6 and tex < 1e-6 and frame < 1e-6


def test_fused_update_equals_separate(oracle, golden):
    # oracle.displace(dt) == update(dt) then displace(0): what the HIP row pass fuses
    p = oracle.EXAMPLE
    ph = golden["phase_60"].copy()
    a = oracle.displace(golden["h0"], ph, p["wavescale"], p["choppiness"], dt=np.float32(1 / 60))
    ph2 = golden["phase_60"].copy()
    oracle.update(ph2, p["wavescale"], np.float32(1 / 60))
    b = oracle.displace(golden["h0"], ph2.copy(), p["wavescale"], p["choppiness"], dt=0.0)
    assert np.array_equal(ph, ph2)
    assert np.array_equal(a, b)


def test_abi_exports_every_declared_symbol():
    from datum_amd import capi

    header = open(os.path.join(ROOT, "include", "datum_ocean_hip.h")).read()
    declared = set(re.findall(r"\b(datum_ocean_[a-z_]+)\s*\(", header))
    declared -= {"datum_ocean_ctx"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_struct_layout():
    from datum_amd import capi

    S = capi.OceanSet
    # byte offsets of src/renderer/ocean.cpp:33-50 under std430
    want = dict(proj=0, invproj=64, camera_real=128, camera_dual=144, plane=160, swelllength=176, swellamplitude=180,
                swellsteepness=184, swellphase=188, swelldirection=192, scale=200, choppiness=204, smoothing=208, size=212)
    for k, off in want.items():
        assert getattr(S, k).offset == off, k
    assert ctypes.sizeof(S) == 216


def test_abi_argument_errors_without_gpu():
    from datum_amd import capi

    lib = capi.load()
    h = capi.P()
    assert lib.datum_ocean_create(ctypes.byref(h), 0, 100, 1) == capi.EINVAL
    assert b"resolution" in lib.datum_ocean_last_error(None)
    assert lib.datum_ocean_create(ctypes.byref(h), 0, 64, 0) == capi.EINVAL
    assert lib.datum_ocean_create(None, 0, 64, 1) == capi.EINVAL
    assert lib.datum_ocean_displace(None) == capi.EINVAL
    assert lib.datum_ocean_destroy(None) == capi.OK
    # the farm's entry points without a handle / with a short id
    buf = (ctypes.c_char * capi.FARM_ID_BYTES)()
    assert lib.datum_ocean_farm_unique_id(buf, 64) == capi.EINVAL
    assert lib.datum_ocean_farm_unique_id(None, capi.FARM_ID_BYTES) == capi.EINVAL
    assert lib.datum_ocean_farm_init(None, buf, capi.FARM_ID_BYTES, 0, 1, capi.PAYLOAD_XYZ32, 2) == capi.EINVAL
    assert lib.datum_ocean_farm_gather(None, None) == capi.EINVAL
    assert lib.datum_ocean_farm_query(None, 0) == capi.EINVAL
    assert lib.datum_ocean_farm_shutdown(None) == capi.EINVAL
    assert capi.ENOTREADY < 0 and capi.ECOMM < 0             # module codes stay out of the hipError_t range (positive)


@pytest.mark.parametrize("N", [64, 256, 512, 1024, 2048, 4096])
def test_map_layout_formula_and_view(N):
    # the documented layout of include/datum_ocean_hip.h (datum_ocean_bind_maps) against capi.map_layers, no GPU
    from datum_amd import capi

    PW, PH, B, TB = capi.map_layout(N)
    assert N % B == 0 and B % PW == 0
    rs = np.random.RandomState(N)
    ys, xs = rs.randint(0, N, 4096), rs.randint(0, N, 4096)
    ys[:4], xs[:4] = [0, 0, N - 1, N - 1], [0, N - 1, 0, N - 1]
    ident = (1 + ys * N + xs).astype(np.float32)                # (exact in fp32 up to 2^24 = 4096^2)
    assert TB == 24
    if True:
        # patches of PW x PH = 16 texels, 384 bytes: 16 x (dx, dy, dz, nx) then 16 x (ny, nz)
        assert PW * PH == 16 and (PW, PH) == ((8, 2) if N <= 256 else (2, 8) if N in (512, 4096) else (4, 4))
        raw = np.zeros(N * N * 6, np.float32)
        patch = (xs // B) * 24 * N * B + ((ys // PH) * (B // PW) + (xs % B) // PW) * 384
        j = (ys % PH) * PW + xs % PW
        a, b = (patch + 16 * j) // 4, (patch + 256 + 8 * j) // 4
        for k in range(4):
            raw[a + k] = ident + 0.25 * k if N < 2048 else ident     # component tags only where fp32 still holds them
        raw[b + 0] = ident
        raw[b + 1] = ident
        view = capi.map_layers(raw, N)
        assert view.shape == (2, N, N, 4) and np.all(view[..., 3] == 0)
        if N < 2048:
            assert np.array_equal(view[0, ys, xs, :3], ident[:, None] + np.array([0, 0.25, 0.5], np.float32))
            assert np.array_equal(view[1, ys, xs, 0], ident + 0.75)
        else:
            assert np.array_equal(view[0, ys, xs, 0], ident)
        assert np.array_equal(view[1, ys, xs, 1], ident) and np.array_equal(view[1, ys, xs, 2], ident)
        assert np.count_nonzero(view[0, ..., 0]) == len(set(zip(ys.tolist(), xs.tolist())))


def test_reference_weights_match_oracle(oracle):
    from datum_amd import capi

    for N in (64, 256, 1024):
        assert np.array_equal(capi.reference_weights(N), oracle.weights(N))


def test_abi_version_is_exported_and_checked(monkeypatch):
    # the shared object has no soname: the version symbol is what lets a consumer refuse a library built from another
    # revision of the header (ENOTREADY changed sign, map_layout gained an argument, texels went from 32 to 24 bytes)
    from datum_amd import capi

    lib = capi.load()
    # the binding's constant, the header and the library agree (the binding does not read the header at run time: ADVICE r05)
    assert lib.datum_ocean_abi_version() == capi.ABI_VERSION == capi.header_abi_version() >= 7

    # a library that reports another version is refused at load time, before any call goes through it
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "ABI_VERSION", lib.datum_ocean_abi_version() + 1)
    with pytest.raises(OSError, match="ABI version"):
        capi.load()
    monkeypatch.setattr(capi, "_lib", lib)


def test_module_keeps_off_the_null_stream():
    # datum_ocean_farm_partition replaces hipStreamNonBlocking streams by CU-masked ones, which synchronise with the legacy null stream
    # (include/datum_ocean_hip.h): the module must not put anything there itself -- every copy, memset and synchronisation of every entry
    # point names a stream.  The one exception is the twiddle upload inside datum_ocean_create, before any stream of the handle has work.
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hits = []
    for name in ("ocean_capi.hip", "ocean_farm.hip", "ocean_gen.hip", "ocean_literal.hip", "ocean_kernels.hip"):
        text = open(os.path.join(root, "datum_amd", "csrc", name)).read()
        for m in re.finditer(r"\b(hipMemcpy|hipMemset|hipMemcpyDtoH|hipMemcpyHtoD|hipMemcpyDtoD|hipDeviceSynchronize)\s*\(", text):
            line = text.count("\n", 0, m.start()) + 1
            hits.append((name, line, m.group(1)))
        # kernels are launched with an explicit stream argument (hipLaunchKernelGGL's fifth) or through launch(), never on stream 0
        for m in re.finditer(r"hipLaunchKernelGGL\(([^;]*?)\);", text, re.S):
            args = m.group(1)
            assert re.search(r",\s*0,\s*(ctx->stream|stream|f->stream)\s*,", args) or "ctx->stream" in args or "stream" in args, (name, args[:80])
    assert [h[2] for h in hits] == ["hipMemcpy"] and hits[0][0] == "ocean_capi.hip", hits

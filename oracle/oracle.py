"""ctypes front end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of ocean_oracle.cpp).  Arrays are numpy float32,
C-contiguous; complex fields are [N, N, 2] (re, im), maps are [2, N, N, 4].
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

F = ctypes.c_float
I = ctypes.c_int
P = ctypes.c_void_p


class OceanSet(ctypes.Structure):
    """Head of the reference's OceanSet SSBO (src/renderer/ocean.cpp:33-50), 216 bytes."""

    _fields_ = [
        ("proj", F * 16),
        ("invproj", F * 16),
        ("camera_real", F * 4),
        ("camera_dual", F * 4),
        ("plane", F * 4),
        ("swelllength", F),
        ("swellamplitude", F),
        ("swellsteepness", F),
        ("swellphase", F),
        ("swelldirection", F * 2),
        ("scale", F),
        ("choppiness", F),
        ("smoothing", F),
        ("size", ctypes.c_uint32),
    ]


assert ctypes.sizeof(OceanSet) == 216


def build(force=False):
    src = os.path.join(_HERE, "ocean_oracle.cpp")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.oracle_dispersion.restype = F
        _lib.oracle_dispersion.argtypes = [F, F]
        _lib.oracle_phillips.restype = F
        _lib.oracle_phillips.argtypes = [F] * 6
        _lib.oracle_lerp.restype = F
        _lib.oracle_lerp.argtypes = [F] * 3
        _lib.oracle_num_threads.restype = I
        _lib.oracle_seed.restype = I
        _lib.oracle_seed.argtypes = [I, ctypes.c_uint32, I, F, F, F, F, F, P, P]
        _lib.oracle_height_from_seed.argtypes = [I, P, F, F, F, F, F, P]
        _lib.oracle_update.argtypes = [I, F, F, P]
        _lib.oracle_update_mt.argtypes = [I, F, F, P]
        _lib.oracle_update_scalars.argtypes = [F, F, F, F, F, F, P, P]
        _lib.oracle_weights.argtypes = [I, P]
        _lib.oracle_weights_reduced.argtypes = [I, P]
        _lib.oracle_sim.argtypes = [I, F, P, P, P, P, P]
        _lib.oracle_fftx.argtypes = [I, P, P]
        _lib.oracle_ffty.argtypes = [I, P, P]
        _lib.oracle_map.argtypes = [I, F, F, P, P, P, P]
        _lib.oracle_displace.argtypes = [I, F, F, F, P, P, P, P, P, I]
        _lib.oracle_camera.argtypes = [F, F, F, F, P, P, P, P]
        _lib.oracle_gen.argtypes = [P, I, P, I, I, P]
        _lib.oracle_indices.argtypes = [I, I, P]
    return _lib


def _p(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# example-ocean parameters (examples/ocean/ocean.cpp:46-50 over the defaults of src/renderer/ocean.h:50-65)
EXAMPLE = dict(
    wavescale=22.0,
    waveamplitude=0.0025,
    windspeed=7.9,
    winddirection=(0.780869, 0.624695),
    choppiness=1.35,
    smoothing=320.0,
    swelllength=40.0,
    swellamplitude=0.8,
    swellsteepness=0.0,
    swellspeed=1.25,
    swelldirection=(0.780869, 0.624695),
    plane=(0.0, 0.0, 1.0, 0.0),
)

# BASELINE/SURVEY section 8(d): cascade c uses wavescale {22, 64, 176, 512}[c]
CASCADE_WAVESCALES = (22.0, 64.0, 176.0, 512.0)


def dispersion(kx, ky):
    return lib().oracle_dispersion(kx, ky)


def phillips(kx, ky, a, v, wx, wy):
    return lib().oracle_phillips(kx, ky, a, v, wx, wy)


def seed(N, rngseed, wavescale=22.0, waveamplitude=0.0025, windspeed=7.9, winddirection=(0.780869, 0.624695), sanitize=True, return_rejected=False):
    """seed_ocean with an explicit mt19937 seed.  sanitize=False is the literal reference behaviour
    (a pair whose 8 polar draws are all rejected becomes NaN); sanitize=True zeroes such pairs."""
    s = np.empty((N, N, 2), np.float32)
    h0 = np.empty((N, N, 2), np.float32)
    rej = lib().oracle_seed(N, rngseed, int(sanitize), wavescale, waveamplitude, windspeed, winddirection[0], winddirection[1], _p(s), _p(h0))
    return (s, h0, rej) if return_rejected else (s, h0)


def height_from_seed(s, wavescale, waveamplitude, windspeed, winddirection):
    s = _f32(s)
    N = s.shape[0]
    h0 = np.empty((N, N, 2), np.float32)
    lib().oracle_height_from_seed(N, _p(s), wavescale, waveamplitude, windspeed, winddirection[0], winddirection[1], _p(h0))
    return h0


def update(phase, wavescale, dt, mt=False):
    """In-place phase advance (src/renderer/ocean.cpp:223-233)."""
    assert phase.dtype == np.float32
    N = phase.shape[0]
    (lib().oracle_update_mt if mt else lib().oracle_update)(N, wavescale, dt, _p(phase))
    return phase


def update_scalars(swellspeed, swelllength, windspeed, winddirection, dt, swellphase, flow):
    sp = F(swellphase)
    fl = (F * 2)(*flow)
    lib().oracle_update_scalars(swellspeed, swelllength, windspeed, winddirection[0], winddirection[1], dt, ctypes.byref(sp), fl)
    return sp.value, (fl[0], fl[1])


def weights(N, reduced=False):
    """Twiddle table of src/renderer/ocean.cpp:686-700.  reduced=True evaluates the same formula with the lane
    index taken modulo the stage period (mathematically identical, accurate at large N: see ocean_oracle.cpp)."""
    stages = int(np.log2(N))
    w = np.empty((N, 2 * stages), np.float32)
    (lib().oracle_weights_reduced if reduced else lib().oracle_weights)(N, _p(w))
    return w


def sim(h0, phase, scale):
    h0, phase = _f32(h0), _f32(phase)
    N = h0.shape[0]
    h, hx, hy = (np.empty((N, N, 2), np.float32) for _ in range(3))
    lib().oracle_sim(N, scale, _p(h0), _p(phase), _p(h), _p(hx), _p(hy))
    return h, hx, hy


def fftx(field, w=None):
    field = _f32(field).copy()
    N = field.shape[0]
    w = weights(N) if w is None else w
    lib().oracle_fftx(N, _p(w), _p(field))
    return field


def ffty(field, w=None):
    field = _f32(field).copy()
    N = field.shape[0]
    w = weights(N) if w is None else w
    lib().oracle_ffty(N, _p(w), _p(field))
    return field


def make_map(h, hx, hy, scale, choppiness):
    h, hx, hy = _f32(h), _f32(hx), _f32(hy)
    N = h.shape[0]
    m = np.empty((2, N, N, 4), np.float32)
    lib().oracle_map(N, scale, choppiness, _p(h), _p(hx), _p(hy), _p(m))
    return m


def displace(h0, phase, wavescale, choppiness, dt=0.0, w=None, mt=False, scratch=None, out=None):
    """update (if dt != 0) -> sim -> fftx -> ffty -> map.  phase is advanced in place."""
    h0 = _f32(h0)
    assert phase.dtype == np.float32 and phase.flags["C_CONTIGUOUS"]
    N = h0.shape[0]
    w = weights(N) if w is None else w
    scratch = np.empty(6 * N * N, np.float32) if scratch is None else scratch
    out = np.empty((2, N, N, 4), np.float32) if out is None else out
    lib().oracle_displace(N, wavescale, choppiness, dt, _p(h0), _p(phase), _p(w), _p(scratch), _p(out), int(mt))
    return out


def camera(fov, aspect, znear, zfar, position, target, up):
    s = OceanSet()
    pos = (F * 3)(*position)
    tgt = (F * 3)(*target)
    upv = (F * 3)(*up)
    lib().oracle_camera(fov, aspect, znear, zfar, pos, tgt, upv, ctypes.byref(s))
    return s


def oceanset(N, position=(0, 0, 8), target=(1, 0, 8), up=(0, 0, 1), params=None, swellphase=0.0,
             fov=60.0 * np.pi / 180.0, aspect=1920.0 / 1080.0, znear=0.1, zfar=24000.0):
    """OceanSet header (src/renderer/ocean.cpp:729-746) for a lookat camera and EXAMPLE overridden by `params`.
    The defaults are the example-ocean camera (examples/ocean/ocean.cpp:33,63; ocean.h:17-18)."""
    p = dict(EXAMPLE)
    if params:
        p.update(params)
    s = camera(fov, aspect, znear, zfar, position, target, up)
    s.plane[:] = p["plane"]
    s.swelllength = p["swelllength"]
    s.swellamplitude = p["swellamplitude"]
    s.swellsteepness = p["swellsteepness"]
    s.swellphase = swellphase
    s.swelldirection[:] = p["swelldirection"]
    s.scale = np.float32(1.0) / np.float32(p["wavescale"])
    s.choppiness = p["choppiness"]
    s.smoothing = np.float32(1.0) / np.float32(p["smoothing"])
    s.size = N
    return s


def example_oceanset(N, params=None, swellphase=0.0):
    """OceanSet header for the example-ocean camera (examples/ocean/ocean.cpp:33,63; ocean.h:17-18)."""
    return oceanset(N, params=params, swellphase=swellphase)


def gen(oceanset, displacementmap, sizex, sizey):
    m = _f32(displacementmap)
    N = m.shape[1]
    v = np.empty((sizey, sizex, 12), np.float32)
    lib().oracle_gen(ctypes.byref(oceanset), N, _p(m), sizex, sizey, _p(v))
    return v


def indices(sizex, sizey):
    idx = np.empty(6 * (sizex - 1) * (sizey - 1), np.uint32)
    lib().oracle_indices(sizex, sizey, _p(idx))
    return idx


def num_threads():
    return lib().oracle_num_threads()

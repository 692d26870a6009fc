"""ctypes view of the C++ host shim (datum_amd/host/ocean.h through host_capi.cpp).

Mirrors the reference's call sites: OceanParams + seed_ocean / lerp_ocean_* / update_ocean on the host,
OceanContext + initialise/prepare/render_ocean_surface over the HIP module.  No compute in Python.
"""

import ctypes
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "lib", "libdatum_ocean_host.so")

F, I, P, U32 = ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32


class Scalars(ctypes.Structure):
    _fields_ = [
        ("plane", F * 4),
        ("swelllength", F), ("swellamplitude", F), ("swellsteepness", F), ("swellspeed", F),
        ("swelldirection", F * 2),
        ("wavescale", F), ("waveamplitude", F), ("windspeed", F),
        ("winddirection", F * 2),
        ("choppiness", F), ("smoothing", F),
        ("swellphase", F),
        ("flow", F * 2),
        ("resolution", I), ("rejectedseeds", I), ("pending", I),
    ]


class CameraDesc(ctypes.Structure):
    _fields_ = [("fov", F), ("aspect", F), ("znear", F), ("zfar", F), ("position", F * 3), ("target", F * 3), ("up", F * 3)]


def example_camera():
    """examples/ocean/ocean.cpp:33,63 with examples/ocean/ocean.h:17-18"""
    c = CameraDesc()
    c.fov = np.float32(60.0) * np.float32(np.pi) / np.float32(180.0)
    c.aspect = np.float32(1920.0) / np.float32(1080.0)
    c.znear, c.zfar = 0.1, 24000.0
    c.position[:] = (0, 0, 8)
    c.target[:] = (1, 0, 8)
    c.up[:] = (0, 0, 1)
    return c


_lib = None


def load():
    global _lib
    if _lib is None:
        capi.load()  # dependency, and fails loudly first
        if not os.path.exists(LIBPATH):
            raise OSError(f"{LIBPATH} not found: build the host shim first (`make` or __graft_entry__.build())")
        lib = ctypes.CDLL(LIBPATH)
        sig = {
            "datum_host_last_error": (ctypes.c_char_p, []),
            "datum_host_params_create": (P, [I]),
            "datum_host_params_destroy": (None, [P]),
            "datum_host_params_clone": (P, [P]),
            "datum_host_params_get": (None, [P, ctypes.POINTER(Scalars)]),
            "datum_host_params_set": (None, [P, ctypes.POINTER(Scalars)]),
            "datum_host_params_set_deviceheight": (None, [P, I]),
            "datum_host_params_set_hostphase": (I, [P, I]),
            "datum_host_params_poke_hostphase": (None, [P, I]),
            "datum_host_release_parked_states": (ctypes.c_size_t, [P, P]),
            "datum_host_parked_states": (I, [P]),
            "datum_host_params_to_pod": (I, [P, P]),
            "datum_host_params_from_pod": (P, [P]),
            "datum_host_pod_bytes": (I, []),
            "datum_host_params_seed": (ctypes.POINTER(F), [P]),
            "datum_host_params_height": (ctypes.POINTER(F), [P]),
            "datum_host_params_phase": (ctypes.POINTER(F), [P]),
            "datum_host_seed_ocean": (None, [P, U32, I]),
            "datum_host_lerp_ocean_swell": (None, [P, F, F, F, F, F, F]),
            "datum_host_lerp_ocean_waves": (None, [P, F, F, F, F, F, F]),
            "datum_host_update_ocean": (I, [P, F]),
            "datum_host_make_oceanset": (None, [ctypes.POINTER(CameraDesc), P, ctypes.POINTER(capi.OceanSet)]),
            "datum_host_twiddle_table": (I, [I, P]),
            "datum_host_context_create": (P, [I, I]),
            "datum_host_context_create_ex": (P, [I, I, I]),
            "datum_host_context_destroy": (None, [P]),
            "datum_host_context_handle": (P, [P]),
            "datum_host_ocean_create": (P, [P, I, I]),
            "datum_host_ocean_release": (None, [P, P]),
            "datum_host_ocean_vertices": (P, [P]),
            "datum_host_ocean_indices": (I, [P, P, P]),
            "datum_host_render_ocean_surface": (I, [P, P, ctypes.POINTER(CameraDesc), P]),
            "datum_host_displace_ocean_surface": (I, [P, P]),
            "datum_host_fetch_ocean_state": (I, [P, P]),
            "datum_host_read_ocean_displacement": (I, [P, P]),
            "datum_host_read_ocean_vertices": (I, [P, P, P]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class OceanParams:
    """OceanParams of datum_amd/host/ocean.h (src/renderer/ocean.h:48-73)."""

    def __init__(self, resolution=64, **tunables):
        self.lib = load()
        self.N = resolution
        self.p = self.lib.datum_host_params_create(resolution)
        if tunables:
            self.set(**tunables)

    def __del__(self):
        if getattr(self, "p", None):
            self.lib.datum_host_params_destroy(self.p)
            self.p = None

    def copy(self):
        """A C++ copy of the OceanParams (the reference's is a POD that game code copies freely)."""
        c = OceanParams.__new__(OceanParams)
        c.lib, c.N = self.lib, self.N
        c.p = self.lib.datum_host_params_clone(self.p)
        return c

    def to_pod(self):
        """The reference's OceanParams byte for byte (OceanParamsPod, 82000 bytes), or None while recorded steps are missing from the host phase."""
        buf = ctypes.create_string_buffer(self.lib.datum_host_pod_bytes())
        rc = self.lib.datum_host_params_to_pod(self.p, buf)
        if rc < 0:
            raise HostError(self.lib.datum_host_last_error().decode())
        return bytes(buf) if rc == 0 else None

    @classmethod
    def from_pod(cls, pod):
        lib = load()
        assert len(pod) == lib.datum_host_pod_bytes()
        c = cls.__new__(cls)
        c.lib, c.N = lib, 64
        c.p = lib.datum_host_params_from_pod(ctypes.create_string_buffer(pod, len(pod)))
        return c

    def scalars(self):
        s = Scalars()
        self.lib.datum_host_params_get(self.p, ctypes.byref(s))
        return s

    def set(self, **kw):
        s = self.scalars()
        for k, v in kw.items():
            cur = getattr(s, k)
            if hasattr(cur, "__len__"):
                cur[:] = v
            else:
                setattr(s, k, v)
        self.lib.datum_host_params_set(self.p, ctypes.byref(s))

    def _arr(self, fn, shape):
        ptr = fn(self.p)
        return np.ctypeslib.as_array(ptr, shape=shape)

    @property
    def seed(self):
        return self._arr(self.lib.datum_host_params_seed, (self.N, self.N, 2))

    @property
    def height(self):
        return self._arr(self.lib.datum_host_params_height, (self.N, self.N, 2))

    @property
    def phase(self):
        return self._arr(self.lib.datum_host_params_phase, (self.N, self.N))

    def set_deviceheight(self, on=True):
        """Extension: lerp_ocean_waves leaves the h0 rebuild to the device (datum_ocean_rebuild_height)."""
        self.lib.datum_host_params_set_deviceheight(self.p, 1 if on else 0)

    def set_hostphase(self, on=True):
        """Extension: update_ocean also advances the host copy of the phase, as the reference does (ocean.cpp:223-233)."""
        if self.lib.datum_host_params_set_hostphase(self.p, 1 if on else 0) != 0:
            raise HostError(self.lib.datum_host_last_error().decode())

    def poke_hostphase(self, on=True):
        """OceanParams::hostphase = on, as C++ code sets the public field (no validation; tests)."""
        self.lib.datum_host_params_poke_hostphase(self.p, 1 if on else 0)

    def seed_ocean(self, rngseed=None):
        self.lib.datum_host_seed_ocean(self.p, 0 if rngseed is None else rngseed, 1 if rngseed is None else 0)

    def lerp_ocean_swell(self, swelllength, swellamplitude, swellspeed, swelldirection, t):
        self.lib.datum_host_lerp_ocean_swell(self.p, swelllength, swellamplitude, swellspeed, swelldirection[0], swelldirection[1], t)

    def lerp_ocean_waves(self, wavescale, waveamplitude, windspeed, winddirection, t):
        self.lib.datum_host_lerp_ocean_waves(self.p, wavescale, waveamplitude, windspeed, winddirection[0], winddirection[1], t)

    def update_ocean(self, dt):
        if self.lib.datum_host_update_ocean(self.p, dt) != 0:
            raise HostError(self.lib.datum_host_last_error().decode())

    def oceanset(self, camera=None):
        out = capi.OceanSet()
        cam = camera or example_camera()
        self.lib.datum_host_make_oceanset(ctypes.byref(cam), self.p, ctypes.byref(out))
        return out


# example-ocean tunables (examples/ocean/ocean.cpp:46-50)
EXAMPLE_TUNABLES = dict(wavescale=22.0, waveamplitude=0.0025, swellamplitude=0.8, windspeed=7.9, smoothing=320.0)


def twiddle_table(N):
    stages = int(np.log2(N))
    w = np.empty((N, 2 * stages), np.float32)
    if load().datum_host_twiddle_table(N, w.ctypes.data_as(P)) != 0:
        raise RuntimeError(load().datum_host_last_error().decode())
    return w


class HostError(RuntimeError):
    pass


class OceanContext:
    """OceanContext after initialise_ocean_context + prepare_ocean_context, with a ResourceManager for Ocean meshes."""

    def __init__(self, resolution=64, device=0, spectrumfp16=False, literaltransform=False, heightfp16=False):
        """spectrumfp16 / literaltransform / heightfp16: the extension flags of the C++ OceanContext (datum_amd/host/ocean.h), set before prepare_ocean_context"""
        self.lib = load()
        self.N = resolution
        self.c = self.lib.datum_host_context_create_ex(device, resolution, (1 if spectrumfp16 else 0) | (2 if literaltransform else 0) | (4 if heightfp16 else 0))
        if not self.c:
            raise HostError(self.lib.datum_host_last_error().decode())
        self.meshes = []

    def close(self):
        if getattr(self, "c", None):
            for m in self.meshes:
                self.lib.datum_host_ocean_release(self.c, m)
            self.meshes = []
            self.lib.datum_host_context_destroy(self.c)
            self.c = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise HostError(self.lib.datum_host_last_error().decode())

    def create_ocean(self, sizex, sizey):
        m = self.lib.datum_host_ocean_create(self.c, sizex, sizey)
        if not m:
            raise HostError(self.lib.datum_host_last_error().decode())
        self.meshes.append(m)
        return (m, sizex, sizey)

    def render_ocean_surface(self, mesh, params, camera=None):
        cam = camera or example_camera()
        self._check(self.lib.datum_host_render_ocean_surface(self.c, mesh[0], ctypes.byref(cam), params.p))

    def displace_ocean_surface(self, params):
        self._check(self.lib.datum_host_displace_ocean_surface(self.c, params.p))

    def release_parked_states(self, keep=None):
        """Free the context's parked device copies (all but `keep`'s); returns the bytes given back."""
        return self.lib.datum_host_release_parked_states(self.c, keep.p if keep is not None else None)

    def parked_states(self):
        return self.lib.datum_host_parked_states(self.c)

    def fetch_ocean_state(self, params):
        self._check(self.lib.datum_host_fetch_ocean_state(self.c, params.p))

    def read_displacement(self):
        out = np.empty((2, self.N, self.N, 4), np.float32)
        self._check(self.lib.datum_host_read_ocean_displacement(self.c, out.ctypes.data_as(P)))
        return out

    def read_vertices(self, mesh):
        out = np.empty((mesh[2], mesh[1], 12), np.float32)
        self._check(self.lib.datum_host_read_ocean_vertices(self.c, mesh[0], out.ctypes.data_as(P)))
        return out

    def read_indices(self, mesh):
        out = np.empty(6 * (mesh[1] - 1) * (mesh[2] - 1), np.uint32)
        rc = self.lib.datum_host_ocean_indices(self.c, mesh[0], out.ctypes.data_as(P))
        if rc != 0:
            raise HostError(f"datum_ocean_device_read failed: {rc}")
        return out

"""ctypes binding of include/datum_ocean_hip.h (libdatum_ocean_hip.so).

No compute happens here and there is no fallback: if the library is missing, import-time loading raises.
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DATUM_OCEAN_HIP_LIB selects another build of the same module (tuning variants); never a different implementation
LIBPATH = os.environ.get("DATUM_OCEAN_HIP_LIB") or os.path.join(_HERE, "lib", "libdatum_ocean_hip.so")

F = ctypes.c_float
I = ctypes.c_int
P = ctypes.c_void_p
D = ctypes.c_double

SUPPORTED = (64, 128, 256, 512, 1024, 2048, 4096)
MAX_CASCADES = 16

OK, EINVAL, ESTATE, ENOMEM, EUNSUPPORTED, ENOTREADY, ECOMM = 0, -1, -2, -3, -4, -5, -6
FARM_ID_BYTES = 128
PAYLOAD_MAPS, PAYLOAD_XYZ32, PAYLOAD_XYZ16 = 0, 1, 2


class OceanSet(ctypes.Structure):
    """datum_ocean_set == head of the reference's OceanSet (src/renderer/ocean.cpp:33-50), 216 bytes."""

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

# every symbol include/datum_ocean_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "datum_ocean_create": (I, [ctypes.POINTER(P), I, I, I]),
    "datum_ocean_destroy": (I, [P]),
    "datum_ocean_set_stream": (I, [P, P, I]),
    "datum_ocean_bind_maps": (I, [P, P, ctypes.c_size_t]),
    "datum_ocean_maps_device": (I, [P, ctypes.POINTER(P), ctypes.POINTER(ctypes.c_size_t)]),
    "datum_ocean_map_layout": (I, [I, ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(I)]),
    "datum_ocean_set_cascade": (I, [P, I, F, F]),
    "datum_ocean_set_spectrum_format": (I, [P, I]),
    "datum_ocean_set_map_store_policy": (I, [P, I]),
    "datum_ocean_map_store_policy": (I, [P, ctypes.POINTER(I), ctypes.POINTER(I)]),
    "datum_ocean_upload_state": (I, [P, I, P, P]),
    "datum_ocean_read_state": (I, [P, I, P]),
    "datum_ocean_state_bytes": (ctypes.c_size_t, [I]),
    "datum_ocean_park_state": (I, [P, I, P, ctypes.c_size_t, ctypes.POINTER(I)]),
    "datum_ocean_resume_state": (I, [P, I, P, ctypes.c_size_t, I]),
    "datum_ocean_upload_seed": (I, [P, I, P]),
    "datum_ocean_rebuild_height": (I, [P, I, F, F, F, F, F]),
    "datum_ocean_read_height": (I, [P, I, P]),
    "datum_ocean_update": (I, [P, F]),
    "datum_ocean_displace": (I, [P]),
    "datum_ocean_gen": (I, [P, I, ctypes.POINTER(OceanSet), I, I, P]),
    "datum_ocean_payload_bytes": (I, [P, I, ctypes.POINTER(ctypes.c_size_t)]),
    "datum_ocean_pack_displacement": (I, [P, I, P, ctypes.c_size_t]),
    "datum_ocean_farm_unique_id": (I, [P, ctypes.c_size_t]),
    "datum_ocean_farm_init": (I, [P, P, ctypes.c_size_t, I, I, I, I]),
    "datum_ocean_farm_shutdown": (I, [P]),
    "datum_ocean_farm_info": (I, [P, ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(I), ctypes.POINTER(I)]),
    "datum_ocean_farm_gather": (I, [P, ctypes.POINTER(I)]),
    "datum_ocean_farm_result": (I, [P, I, P, I, ctypes.POINTER(P), ctypes.POINTER(ctypes.c_size_t)]),
    "datum_ocean_farm_release": (I, [P, I, P, I]),
    "datum_ocean_farm_query": (I, [P, I]),
    "datum_ocean_farm_wait": (I, [P, I, ctypes.POINTER(F)]),
    "datum_ocean_farm_partition": (I, [P, I]),
    "datum_ocean_own_stream": (I, [P, ctypes.POINTER(P)]),
    "datum_ocean_read_maps": (I, [P, I, P]),
    "datum_ocean_sync": (I, [P]),
    "datum_ocean_wait_event": (I, [P, P]),
    "datum_ocean_signal": (I, [P, ctypes.POINTER(P)]),
    "datum_ocean_on_complete": (I, [P, ctypes.CFUNCTYPE(None, ctypes.c_void_p), P]),
    "datum_ocean_query": (I, [P]),
    "datum_ocean_import_memory_fd": (I, [P, I, ctypes.c_size_t, ctypes.POINTER(P)]),
    "datum_ocean_release_memory": (I, [P, P]),
    "datum_ocean_import_semaphore_fd": (I, [P, I, ctypes.POINTER(P)]),
    "datum_ocean_release_semaphore": (I, [P, P]),
    "datum_ocean_signal_external": (I, [P, P]),
    "datum_ocean_wait_external": (I, [P, P]),
    "datum_ocean_device_alloc": (I, [P, ctypes.c_size_t, ctypes.POINTER(P)]),
    "datum_ocean_device_free": (I, [P, P]),
    "datum_ocean_device_write": (I, [P, P, P, ctypes.c_size_t]),
    "datum_ocean_device_read": (I, [P, P, P, ctypes.c_size_t]),
    "datum_ocean_last_error": (ctypes.c_char_p, [P]),
    "datum_ocean_reference_weights": (I, [I, P]),
    "datum_ocean_debug_sim": (I, [P, I, P, P, P]),
    "datum_ocean_debug_rowpass": (I, [P, I, P, P]),
    "datum_ocean_profile_begin": (I, [P, I, I]),
    "datum_ocean_profile_end": (I, [P, ctypes.POINTER(D), ctypes.POINTER(D), ctypes.POINTER(I)]),
    "datum_ocean_algorithmic_bytes": (I, [P, ctypes.POINTER(D), ctypes.POINTER(D)]),
    "datum_ocean_abi_version": (I, []),
    "datum_ocean_set_literal_transform": (I, [P, I]),
    "datum_ocean_export_maps": (I, [P, I, P, ctypes.c_size_t]),
    "datum_ocean_farm_stream_flags": (I, [P, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]),
    "datum_ocean_set_cascade_group": (I, [P, I]),
    "datum_ocean_cascade_group": (I, [P, ctypes.POINTER(I), ctypes.POINTER(I)]),
}


# DATUM_OCEAN_ABI_VERSION of include/datum_ocean_hip.h as SYMBOLS above was written against it.  A constant, not a read of the header: an
# installed or copied package has no include/ beside it (tests/test_golden_and_abi.py asserts that the two agree in the source tree).
ABI_VERSION = 8

# datum_ocean_set_spectrum_format's values (include/datum_ocean_hip.h)
SPECTRUM_FORMATS = {"fp32": 0, "fp16": 1, "fp16h0": 2}

# datum_ocean_set_map_store_policy's values
MAP_STORE_POLICIES = {"auto": 0, "written through": 1, "streamed": 2}


def header_abi_version():
    """DATUM_OCEAN_ABI_VERSION as include/datum_ocean_hip.h states it (source tree only: the tests compare it with ABI_VERSION)."""
    import re

    text = open(os.path.join(os.path.dirname(_HERE), "include", "datum_ocean_hip.h")).read()
    return int(re.search(r"#define\s+DATUM_OCEAN_ABI_VERSION\s+(\d+)", text).group(1))

_lib = None


def load():
    """Load the HIP module.  Raises OSError when it is not built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise OSError(
                f"{LIBPATH} not found: build the HIP module first (`make` or __graft_entry__.build()); "
                "datum_amd has no CPU fallback"
            )
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 + libhsa-runtime64 and the
        # module links the same SONAME from /opt/rocm.  Whichever is mapped first serves both; torch cannot
        # initialise on top of the system pair, so when torch is importable it goes first (it is also what the
        # tests and bench.py use for device buffers and streams, which must come from the same runtime).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIBPATH)
        # no soname: a library built from another revision of the header (DATUM_OCEAN_HIP_LIB swaps builds freely) is refused
        # before any call goes through a signature it may not have
        try:
            lib.datum_ocean_abi_version.restype = I
            lib.datum_ocean_abi_version.argtypes = []
            have = lib.datum_ocean_abi_version()
        except AttributeError:
            have = None
        want = ABI_VERSION
        if have != want:
            raise OSError(f"{LIBPATH} reports ABI version {have}, this binding is written against version {want} of include/datum_ocean_hip.h: rebuild the HIP module (`make`)")
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class OceanError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"datum_ocean error {code}: {message}")
        self.code = code


def _ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P)


class Ocean:
    """One handle of the C ABI (a device, a resolution, `cascades` independent grids)."""

    def __init__(self, resolution, cascades=1, device=0):
        self.lib = load()
        self.N = resolution
        self.cascades = cascades
        self.device = device
        h = P()
        rc = self.lib.datum_ocean_create(ctypes.byref(h), device, resolution, cascades)
        if rc != 0:
            raise OceanError(rc, self.lib.datum_ocean_last_error(None).decode())
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise OceanError(rc, self.lib.datum_ocean_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.datum_ocean_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_ptr):
        """stream_ptr: a hipStream_t as an integer (0 = HIP's default stream); None = the handle's own stream."""
        if stream_ptr is None:
            self._check(self.lib.datum_ocean_set_stream(self.h, None, 1))
        else:
            self._check(self.lib.datum_ocean_set_stream(self.h, P(stream_ptr), 0))

    def bind_maps(self, device_ptr, nbytes):
        self._check(self.lib.datum_ocean_bind_maps(self.h, P(device_ptr) if device_ptr else None, nbytes))

    def maps_device(self):
        p = P()
        n = ctypes.c_size_t()
        self._check(self.lib.datum_ocean_maps_device(self.h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def set_cascade(self, cascade, wavescale, choppiness):
        self._check(self.lib.datum_ocean_set_cascade(self.h, cascade, wavescale, choppiness))

    def upload_state(self, cascade, h0, phase=None):
        h0 = np.ascontiguousarray(h0, np.float32)
        assert h0.size == 2 * self.N * self.N
        if phase is not None:
            phase = np.ascontiguousarray(phase, np.float32)
            assert phase.size == self.N * self.N
        self._check(self.lib.datum_ocean_upload_state(self.h, cascade, _ptr(h0), _ptr(phase) if phase is not None else None))

    def upload_seed(self, cascade, seed):
        seed = np.ascontiguousarray(seed, np.float32)
        assert seed.size == 2 * self.N * self.N
        self._check(self.lib.datum_ocean_upload_seed(self.h, cascade, _ptr(seed)))

    def rebuild_height(self, cascade, wavescale, waveamplitude, windspeed, winddirection):
        self._check(self.lib.datum_ocean_rebuild_height(self.h, cascade, wavescale, waveamplitude, windspeed, winddirection[0], winddirection[1]))

    def read_height(self, cascade):
        out = np.empty((self.N, self.N, 2), np.float32)
        self._check(self.lib.datum_ocean_read_height(self.h, cascade, _ptr(out)))
        return out

    def read_state(self, cascade):
        out = np.empty((self.N, self.N), np.float32)
        self._check(self.lib.datum_ocean_read_state(self.h, cascade, _ptr(out)))
        return out

    def update(self, dt):
        self._check(self.lib.datum_ocean_update(self.h, dt))

    def displace(self):
        self._check(self.lib.datum_ocean_displace(self.h))

    def gen(self, cascade, oceanset, sizex, sizey, vertices_device_ptr):
        self._check(self.lib.datum_ocean_gen(self.h, cascade, ctypes.byref(oceanset), sizex, sizey, P(vertices_device_ptr)))

    def payload_bytes(self, fmt):
        n = ctypes.c_size_t()
        self._check(self.lib.datum_ocean_payload_bytes(self.h, fmt, ctypes.byref(n)))
        return n.value

    def pack_displacement(self, fmt, device_ptr, nbytes):
        """Enqueue the packing of every cascade's displacement into caller-owned device memory (all-gather payload)."""
        self._check(self.lib.datum_ocean_pack_displacement(self.h, fmt, P(device_ptr), nbytes))

    # -- the tile farm (include/datum_ocean_hip.h: datum_ocean_farm_*): the RCCL all-gather lives in the module --------

    def farm_init(self, unique_id, rank, world, fmt, slots=2):
        """Join the farm's communicator (collective over the ranks).  unique_id: the 128 bytes of farm_unique_id() on rank 0."""
        assert len(unique_id) == FARM_ID_BYTES
        buf = (ctypes.c_char * FARM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.datum_ocean_farm_init(self.h, buf, FARM_ID_BYTES, rank, world, fmt, slots))

    def farm_shutdown(self):
        self._check(self.lib.datum_ocean_farm_shutdown(self.h))

    def farm_info(self):
        rank, world, fmt, slots, ver = I(), I(), I(), I(), I()
        nbytes = ctypes.c_size_t()
        self._check(self.lib.datum_ocean_farm_info(self.h, ctypes.byref(rank), ctypes.byref(world), ctypes.byref(fmt), ctypes.byref(nbytes), ctypes.byref(slots), ctypes.byref(ver)))
        return dict(rank=rank.value, world=world.value, format=fmt.value, payload_bytes=nbytes.value, slots=slots.value, rccl_version=ver.value)

    def farm_gather(self):
        """Enqueue pack + all-gather of the displacement field as of the last displace; returns the slot.  Does not block."""
        slot = I()
        self._check(self.lib.datum_ocean_farm_gather(self.h, ctypes.byref(slot)))
        return slot.value

    def farm_result(self, slot, stream_ptr=None):
        """(device pointer, bytes) of the slot's gathered block; `stream_ptr` (None: the handle's stream) waits for the collective."""
        p, n = P(), ctypes.c_size_t()
        self._check(self.lib.datum_ocean_farm_result(self.h, slot, P(stream_ptr) if stream_ptr else None, 1 if stream_ptr is None else 0, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def farm_release(self, slot, stream_ptr=None):
        self._check(self.lib.datum_ocean_farm_release(self.h, slot, P(stream_ptr) if stream_ptr else None, 1 if stream_ptr is None else 0))

    def farm_query(self, slot):
        rc = self.lib.datum_ocean_farm_query(self.h, slot)
        if rc == ENOTREADY:
            return False
        self._check(rc)
        return True

    def farm_partition(self, comm_cus):
        """The communication stream on `comm_cus` compute units of its own (comm_cus / 8 per XCD), the handle's own stream on the others; 0 undoes it."""
        self._check(self.lib.datum_ocean_farm_partition(self.h, comm_cus))

    def own_stream(self):
        """The handle's own hipStream_t as an integer (torch.cuda.ExternalStream takes it)."""
        p = P()
        self._check(self.lib.datum_ocean_own_stream(self.h, ctypes.byref(p)))
        return p.value

    def farm_wait(self, slot):
        """Host wait for the slot's collective; returns its duration on the communication stream in ms."""
        ms = F()
        self._check(self.lib.datum_ocean_farm_wait(self.h, slot, ctypes.byref(ms)))
        return ms.value

    def import_memory_fd(self, fd, nbytes):
        """Device pointer over memory another API exported as a POSIX fd (Vulkan external memory, opaque fd)."""
        p = P()
        self._check(self.lib.datum_ocean_import_memory_fd(self.h, fd, nbytes, ctypes.byref(p)))
        return p.value

    def release_memory(self, device_ptr):
        self._check(self.lib.datum_ocean_release_memory(self.h, P(device_ptr)))

    def signal(self):
        """Record the handle's completion event behind everything enqueued so far; returns the hipEvent_t."""
        p = P()
        self._check(self.lib.datum_ocean_signal(self.h, ctypes.byref(p)))
        return p.value

    def wait_event(self, hip_event):
        self._check(self.lib.datum_ocean_wait_event(self.h, P(hip_event)))

    def query(self):
        """True once everything enqueued before the last signal() has finished; never blocks."""
        rc = self.lib.datum_ocean_query(self.h)
        if rc == ENOTREADY:
            return False
        self._check(rc)
        return True

    def on_complete(self, fn):
        """`fn()` on a runtime thread once everything enqueued so far has finished.  The ctypes trampoline is kept alive on the handle."""
        cb = ctypes.CFUNCTYPE(None, ctypes.c_void_p)(lambda _user: fn())
        self._callbacks = getattr(self, "_callbacks", []) + [cb]
        self._check(self.lib.datum_ocean_on_complete(self.h, cb, None))

    def import_semaphore_fd(self, fd):
        p = P()
        self._check(self.lib.datum_ocean_import_semaphore_fd(self.h, fd, ctypes.byref(p)))
        return p.value

    def state_bytes(self):
        return self.lib.datum_ocean_state_bytes(self.N)

    def park_state(self, cascade, device_ptr, nbytes):
        """h0 and the phase as advanced so far into caller-owned device memory (device to device); returns the flags to hand back."""
        flags = I()
        self._check(self.lib.datum_ocean_park_state(self.h, cascade, P(device_ptr), nbytes, ctypes.byref(flags)))
        return flags.value

    def resume_state(self, cascade, device_ptr, nbytes, flags):
        self._check(self.lib.datum_ocean_resume_state(self.h, cascade, P(device_ptr), nbytes, flags))

    def read_maps(self, cascade):
        out = np.empty((2, self.N, self.N, 4), np.float32)
        self._check(self.lib.datum_ocean_read_maps(self.h, cascade, _ptr(out)))
        return out

    def sync(self):
        self._check(self.lib.datum_ocean_sync(self.h))

    def debug_sim(self, cascade):
        h, hx, hy = (np.empty((self.N, self.N, 2), np.float32) for _ in range(3))
        self._check(self.lib.datum_ocean_debug_sim(self.h, cascade, _ptr(h), _ptr(hx), _ptr(hy)))
        return h, hx, hy

    def set_spectrum_format(self, fp16):
        """Work spectrum between the passes as IEEE halves (True / "fp16") or fp32 (False / "fp32", default); "fp16h0": the halves and h0
        read as halves too (DATUM_OCEAN_SPECTRUM_FP16_H0)."""
        code = SPECTRUM_FORMATS[fp16] if isinstance(fp16, str) else (1 if fp16 else 0)
        self._check(self.lib.datum_ocean_set_spectrum_format(self.h, code))

    def debug_rowpass(self, cascade):
        c, d = (np.empty((self.N, self.N, 2), np.float32) for _ in range(2))
        self._check(self.lib.datum_ocean_debug_rowpass(self.h, cascade, _ptr(c), _ptr(d)))
        return c, d

    def profile_begin(self, max_samples, stride=1):
        self._check(self.lib.datum_ocean_profile_begin(self.h, max_samples, stride))

    def profile_end(self):
        row, col, n = D(), D(), I()
        self._check(self.lib.datum_ocean_profile_end(self.h, ctypes.byref(row), ctypes.byref(col), ctypes.byref(n)))
        return row.value, col.value, n.value

    def set_literal_transform(self, on):
        """validation mode: displace through the reference's radix-2 transforms and literal twiddle table (datum_ocean_set_literal_transform)"""
        self._check(self.lib.datum_ocean_set_literal_transform(self.h, 1 if on else 0))

    def farm_stream_flags(self):
        """hipStreamGetFlags of (communication stream, own stream): 0 = hipStreamDefault (synchronises with the null stream), 1 = hipStreamNonBlocking"""
        a, b = ctypes.c_uint(), ctypes.c_uint()
        self._check(self.lib.datum_ocean_farm_stream_flags(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def set_cascade_group(self, cascades_per_launch):
        """cascades per launch of the two passes; 0 = sized to the Infinity Cache (datum_ocean_set_cascade_group)"""
        self._check(self.lib.datum_ocean_set_cascade_group(self.h, cascades_per_launch))

    def cascade_group(self):
        """(cascades per launch, launches per pass and displace call)"""
        g, n = I(), I()
        self._check(self.lib.datum_ocean_cascade_group(self.h, ctypes.byref(g), ctypes.byref(n)))
        return g.value, n.value

    def set_map_store_policy(self, policy):
        """how the column pass stores the maps: "auto" (default), "written through", "streamed" (datum_ocean_set_map_store_policy)"""
        self._check(self.lib.datum_ocean_set_map_store_policy(self.h, MAP_STORE_POLICIES[policy] if isinstance(policy, str) else int(policy)))

    def map_store_policy(self):
        """(the policy set, whether the next displace streams the maps)"""
        p, s = I(), I()
        self._check(self.lib.datum_ocean_map_store_policy(self.h, ctypes.byref(p), ctypes.byref(s)))
        return {v: k for k, v in MAP_STORE_POLICIES.items()}[p.value], bool(s.value)

    def export_maps(self, cascade, device_ptr, nbytes):
        """the cascade's maps as the reference's [layer][y][x][4] RGBA32F image, into DEVICE memory (datum_ocean_export_maps)"""
        self._check(self.lib.datum_ocean_export_maps(self.h, cascade, ctypes.c_void_p(device_ptr), nbytes))

    def algorithmic_bytes(self):
        row, col = D(), D()
        self._check(self.lib.datum_ocean_algorithmic_bytes(self.h, ctypes.byref(row), ctypes.byref(col)))
        return row.value, col.value


def farm_unique_id():
    """The 128-byte id of a new farm communicator (rank 0 makes it, every rank passes it to Ocean.farm_init)."""
    buf = (ctypes.c_char * FARM_ID_BYTES)()
    rc = load().datum_ocean_farm_unique_id(buf, FARM_ID_BYTES)
    if rc != 0:
        raise OceanError(rc, load().datum_ocean_last_error(None).decode())
    return bytes(buf)


def map_layout(N):
    """(PW, PH, B, texel_bytes) of the device map layout at resolution N (include/datum_ocean_hip.h: datum_ocean_bind_maps):
    patches of PW x PH = 16 texels of 24 bytes, bands of B columns."""
    gx, gy, b, tb = I(), I(), I(), I()
    rc = load().datum_ocean_map_layout(N, ctypes.byref(gx), ctypes.byref(gy), ctypes.byref(b), ctypes.byref(tb))
    if rc != 0:
        raise OceanError(rc, load().datum_ocean_last_error(None).decode())
    return gx.value, gy.value, b.value, tb.value


def map_block_floats(N):
    """floats of one cascade's DEVICE map block"""
    return N * N * map_layout(N)[3] // 4


def map_layers(raw, N):
    """One cascade's DEVICE map block (map_block_floats(N) floats as the kernels lay them out, include/datum_ocean_hip.h) as
    the reference's logical image [layer][y][x][4] (.w = 0).  Works on numpy arrays and torch tensors alike.
    Bands of B columns, patches of PW x PH texels, 16 x (dx, dy, dz, nx) then 16 x (ny, nz) per patch."""
    GX, GY, B, TB = map_layout(N)
    assert TB == 24
    torchlike = hasattr(raw, "permute")
    v = raw.reshape(N // B, N // GY, B // GX, 96)                  # [band][y / PH][patch][96 floats]
    a = v[..., :64].reshape(N // B, N // GY, B // GX, GY, GX, 4)   # (dx, dy, dz, nx) per texel
    b = v[..., 64:].reshape(N // B, N // GY, B // GX, GY, GX, 2)   # (ny, nz) per texel
    order = (1, 3, 0, 2, 4, 5)                                     # -> [y / PH][y % PH][band][patch][x % PW][component]
    if torchlike:
        import torch

        a, b = a.permute(*order).reshape(N, N, 4), b.permute(*order).reshape(N, N, 2)
        z = torch.zeros_like(a[..., :1])
        return torch.stack([torch.cat([a[..., :3], z], -1), torch.cat([a[..., 3:], b, z], -1)])
    a, b = a.transpose(*order).reshape(N, N, 4), b.transpose(*order).reshape(N, N, 2)
    z = np.zeros_like(a[..., :1])
    return np.stack([np.concatenate([a[..., :3], z], -1), np.concatenate([a[..., 3:], b, z], -1)])


def reference_weights(N):
    stages = int(np.log2(N))
    w = np.empty((N, 2 * stages), np.float32)
    rc = load().datum_ocean_reference_weights(N, _ptr(w))
    if rc != 0:
        raise OceanError(rc, load().datum_ocean_last_error(None).decode())
    return w

"""GPU tests of the Vulkan <-> HIP interop entry points (include/datum_ocean_hip.h: import_memory_fd, import_semaphore_fd,
signal / wait_external) -- the HIP half of SURVEY.md 8(f) rank 1.  No Vulkan loader exists on the box, so the exporting
side is played by tests/gpu/extmem_helper.hip: a HIP virtual-memory allocation exported as a POSIX file descriptor, which
on amdgpu is the same dma-buf descriptor VK_KHR_external_memory_fd hands out for OPAQUE_FD."""

import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = np.float32(1.0 / 60.0)
HERE = os.path.dirname(os.path.abspath(__file__))


class ExtMem(ctypes.Structure):
    _fields_ = [("handle", ctypes.c_void_p), ("va", ctypes.c_void_p), ("bytes", ctypes.c_size_t)]


@pytest.fixture(scope="module")
def capi():
    from datum_amd import capi as c

    c.load()
    return c


@pytest.fixture(scope="module")
def helper(capi):
    lib = ctypes.CDLL(os.path.join(HERE, "gpu", "libextmem_helper.so"))
    lib.extmem_create.argtypes = [ctypes.c_size_t, ctypes.POINTER(ExtMem), ctypes.POINTER(ctypes.c_int)]
    lib.extmem_read.argtypes = [ctypes.POINTER(ExtMem), ctypes.c_void_p, ctypes.c_size_t]
    lib.extmem_destroy.argtypes = [ctypes.POINTER(ExtMem)]
    return lib


def test_gen_into_imported_vertex_buffer(capi, helper, oracle):
    # the renderer's vertex buffer (ocean.cpp:270) exported as an fd, imported here, written by ocean.gen through the
    # imported pointer, read back by the exporter through its own mapping: the bytes the draw call would see
    N, sx, sy = 64, 64, 48
    p = oracle.EXAMPLE
    _, h0 = oracle.seed(N, 1000, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"])
    s = oracle.example_oceanset(N, swellphase=0.4)
    hs = capi.OceanSet.from_buffer_copy(bytes(s))
    nbytes = sx * sy * 48
    mem, fd = ExtMem(), ctypes.c_int(-1)
    assert helper.extmem_create(nbytes, ctypes.byref(mem), ctypes.byref(fd)) == 0
    try:
        with capi.Ocean(N, 1) as oc:
            ptr = oc.import_memory_fd(fd.value, mem.bytes)
            assert ptr and ptr != mem.va                      # a second mapping of the same memory
            oc.set_cascade(0, p["wavescale"], p["choppiness"])
            oc.upload_state(0, h0)
            oc.update(DT)
            oc.displace()
            oc.gen(0, hs, sx, sy, ptr)
            oc.sync()
            maps = oc.read_maps(0)
            got = np.empty((sy, sx, 12), np.float32)
            assert helper.extmem_read(ctypes.byref(mem), got.ctypes.data_as(ctypes.c_void_p), nbytes) == 0
            want = oracle.gen(s, maps, sx, sy)
            assert (np.abs(got[..., 0:3] - want[..., 0:3]) / (1 + np.abs(want[..., 0:3]))).max() < 2e-4
            assert np.abs(got[..., 5:11] - want[..., 5:11]).max() < 2e-4
            assert np.all(got[..., 11] == -1)
            # released explicitly: the pointer is no longer the handle's, a second release is refused
            oc.release_memory(ptr)
            with pytest.raises(capi.OceanError) as e:
                oc.release_memory(ptr)
            assert e.value.code == capi.EINVAL
    finally:
        assert helper.extmem_destroy(ctypes.byref(mem)) == 0


def test_maps_in_imported_memory_and_teardown(capi, helper, oracle):
    # the displacement map in imported memory (datum_ocean_bind_maps); the handle's destruction releases the import
    N = 128
    p = oracle.EXAMPLE
    _, h0 = oracle.seed(N, 1000, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"])
    nbytes = 2 * N * N * 16
    mem, fd = ExtMem(), ctypes.c_int(-1)
    assert helper.extmem_create(nbytes, ctypes.byref(mem), ctypes.byref(fd)) == 0
    try:
        oc = capi.Ocean(N, 1)
        ptr = oc.import_memory_fd(fd.value, mem.bytes)
        oc.set_cascade(0, p["wavescale"], p["choppiness"])
        oc.upload_state(0, h0)
        oc.bind_maps(ptr, nbytes)
        oc.update(DT)
        oc.displace()
        oc.sync()
        own = oc.read_maps(0)
        raw = np.empty(2 * N * N * 4, np.float32)
        assert helper.extmem_read(ctypes.byref(mem), raw.ctypes.data_as(ctypes.c_void_p), nbytes) == 0
        assert np.array_equal(capi.map_layers(raw[:capi.map_block_floats(N)], N), own)
        oc.close()                                        # imports still held: destroy releases them
    finally:
        assert helper.extmem_destroy(ctypes.byref(mem)) == 0


def test_import_errors(capi):
    with capi.Ocean(64, 1) as oc:
        with pytest.raises(capi.OceanError) as e:
            oc.import_memory_fd(-1, 4096)
        assert e.value.code == capi.EINVAL
        r, w = os.pipe()                                  # a descriptor that is not device memory
        try:
            with pytest.raises(capi.OceanError) as e:
                oc.import_memory_fd(r, 4096)
            assert e.value.code > 0                       # the runtime's own error code comes through
        finally:
            os.close(r)
            os.close(w)
        with pytest.raises(capi.OceanError) as e:
            oc.import_memory_fd(0, 0)
        assert e.value.code == capi.EINVAL
        with pytest.raises(capi.OceanError) as e:
            oc.import_semaphore_fd(-1)
        assert e.value.code == capi.EINVAL
        r, w = os.pipe()
        try:
            with pytest.raises(capi.OceanError) as e:
                oc.import_semaphore_fd(r)                 # not a semaphore: refused, not a crash
            # ROCm 7.0.x has no external semaphores at all (profiles/r02_external_memory_probe.txt): that has its own
            # documented code, and the message names the host bridge; a runtime that has them refuses the pipe itself
            assert e.value.code == capi.EUNSUPPORTED or e.value.code > 0
            if e.value.code == capi.EUNSUPPORTED:
                assert "datum_ocean_on_complete" in str(e.value)
        finally:
            os.close(r)
            os.close(w)
        # a pointer / semaphore the handle did not import
        for fn in (oc.lib.datum_ocean_signal_external, oc.lib.datum_ocean_wait_external, oc.lib.datum_ocean_release_semaphore, oc.lib.datum_ocean_release_memory):
            assert fn(oc.h, ctypes.c_void_p(0x1000)) == capi.EINVAL
        oc.update(DT)                                     # the handle is still usable


def _hip():
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    return lib


def test_rendercomplete_bridge_orders_a_second_stream(capi, helper, oracle):
    # `rendercomplete` without external semaphores (fallback): the consumer -- here a second stream that
    # copies the imported vertex buffer, standing in for the draw -- waits on the event of datum_ocean_signal and must see
    # every vertex of the gen that was enqueued before it; and the other way round, a producer on the second stream (a
    # clear of the vertex buffer: the renderer re-using it) is waited for by the handle through datum_ocean_wait_event
    # before gen writes.  Repeated so that a missing edge would show.
    import torch

    hip = _hip()
    D2D = 3
    N, sx, sy = 64, 1024, 512
    p = oracle.EXAMPLE
    _, h0 = oracle.seed(N, 1000, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"])
    s = oracle.example_oceanset(N, swellphase=0.4)
    hs = capi.OceanSet.from_buffer_copy(bytes(s))
    nbytes = sx * sy * 48
    mem, fd = ExtMem(), ctypes.c_int(-1)
    assert helper.extmem_create(nbytes, ctypes.byref(mem), ctypes.byref(fd)) == 0
    try:
        with capi.Ocean(N, 1) as oc:
            ptr = oc.import_memory_fd(fd.value, mem.bytes)
            oc.set_cascade(0, p["wavescale"], p["choppiness"])
            oc.upload_state(0, h0)
            oc.update(DT)
            oc.displace()
            oc.sync()
            maps = oc.read_maps(0)
            want = oracle.gen(s, maps, sx, sy)
            other = torch.cuda.Stream()
            seen = torch.zeros(sx * sy * 12, dtype=torch.float32, device="cuda:0")
            cleared = torch.cuda.Event()
            for _ in range(5):
                # renderer side: clears the buffer on ITS stream, then the ocean may write
                assert hip.hipMemsetAsync(ptr, 0xFF, nbytes, other.cuda_stream) == 0
                cleared.record(other)
                oc.wait_event(cleared.cuda_event)
                oc.gen(0, hs, sx, sy, ptr)
                done = oc.signal()
                # renderer side: waits for rendercomplete, then reads the vertices
                assert hip.hipStreamWaitEvent(ctypes.c_void_p(other.cuda_stream), ctypes.c_void_p(done), 0) == 0
                assert hip.hipMemcpyAsync(seen.data_ptr(), ptr, nbytes, D2D, other.cuda_stream) == 0
                other.synchronize()
                got = seen.cpu().numpy().reshape(sy, sx, 12)
                assert np.isfinite(got).all()              # 0xFFFFFFFF is a NaN: a vertex gen had not written yet, or one cleared after
                assert (np.abs(got[..., 0:3] - want[..., 0:3]) / (1 + np.abs(want[..., 0:3]))).max() < 2e-4
            oc.sync()
    finally:
        assert helper.extmem_destroy(ctypes.byref(mem)) == 0


def test_rendercomplete_on_the_host(capi, oracle):
    # the host half of the bridge: datum_ocean_on_complete fires once the frame's work has finished (the integrator's
    # callback then signals the VkSemaphore), datum_ocean_query polls without blocking; neither synchronises the stream
    import threading
    import time

    import torch

    N, sx, sy = 256, 1024, 1024
    p = oracle.EXAMPLE
    _, h0 = oracle.seed(N, 1000, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"])
    hs = capi.OceanSet.from_buffer_copy(bytes(oracle.example_oceanset(N)))
    verts = torch.full((sx * sy * 12,), float("nan"), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    with capi.Ocean(N, 1) as oc:
        with pytest.raises(capi.OceanError) as e:
            oc.query()                                    # nothing signalled yet
        assert e.value.code == capi.ESTATE
        oc.set_cascade(0, p["wavescale"], p["choppiness"])
        oc.upload_state(0, h0)
        fired = threading.Event()
        order = []
        for _ in range(50):
            oc.update(DT)
            oc.displace()
        oc.gen(0, hs, sx, sy, verts.data_ptr())
        oc.on_complete(lambda: (order.append("callback"), fired.set()))
        oc.signal()
        polls = 0
        t0 = time.perf_counter()
        while not oc.query():
            polls += 1
            assert time.perf_counter() - t0 < 30
        order.append("query")
        assert fired.wait(30)
        # everything enqueued before the callback is visible once it has fired: read on another stream, no sync of the handle's
        other = torch.cuda.Stream()
        with torch.cuda.stream(other):
            host = verts.to("cpu", non_blocking=False)
        assert torch.isfinite(host).all()
        oc.sync()


def test_release_memory_with_maps_bound_inside_the_block(capi, helper, oracle):
    # a VkBuffer commonly sits at an OFFSET inside its VkDeviceMemory: the maps are bound somewhere inside the imported
    # block, then the block is released by its base pointer -- the handle must fall back to its own map buffer instead of
    # writing through a pointer into unmapped memory
    N = 64
    p = oracle.EXAMPLE
    _, h0 = oracle.seed(N, 1000, p["wavescale"], p["waveamplitude"], p["windspeed"], p["winddirection"])
    nbytes = 2 * N * N * 16
    offset = 4096
    mem, fd = ExtMem(), ctypes.c_int(-1)
    assert helper.extmem_create(nbytes + offset, ctypes.byref(mem), ctypes.byref(fd)) == 0
    try:
        with capi.Ocean(N, 1) as oc:
            ptr = oc.import_memory_fd(fd.value, mem.bytes)
            oc.set_cascade(0, p["wavescale"], p["choppiness"])
            oc.upload_state(0, h0)
            oc.bind_maps(ptr + offset, nbytes)
            oc.update(DT)
            oc.displace()
            inside = oc.read_maps(0)
            oc.release_memory(ptr)                       # the maps were bound at ptr + offset
            oc.displace()                                # ... and now land in the handle's own buffer: no fault
            oc.sync()
            assert np.array_equal(oc.read_maps(0), inside)
    finally:
        assert helper.extmem_destroy(ctypes.byref(mem)) == 0

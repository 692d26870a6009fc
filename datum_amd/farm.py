"""Tile / cascade farm across the GPUs of one node.

Cascades and tiles are independent (OceanParams, N x N grid) problems -- own seed, no halo, each grid is periodic
(data/ocean.map.comp:58) -- so rank r simply owns the global grids [r*C, (r+1)*C) and the displacement step needs no
collective.  The one exchange north_star asks for, "a single RCCL all-gather over xGMI to reassemble the displacement
field", is ONE all_gather_into_tensor per batch of steps.  What is gathered (the payload) and when:

  * payload: every rank packs its tiles' displacement into a flat buffer (datum_ocean_pack_displacement,
    include/datum_ocean_hip.h): "xyz32" = (dx, dy, dz) as floats, 12 B per point, exact -- the displacement FIELD
    north_star names; "xyz16" = the same as halves, 8 B per point; "maps" = both map layers as they lie in memory,
    24 B per point (capi.map_layout).  xGMI is point to point (7 links x ~153 GB/s per GPU): at 8 ranks a 1024^2 x 4 block is 0.35 GB
    received per rank as xyz32 against 0.94 GB as maps.
  * overlap: the payload is double-buffered and the collective of batch k runs on a second stream while the kernels
    of batch k + 1 run on the compute stream (TileGather below); event-ordered, no host synchronisation.

This module holds the index arithmetic and the stream / buffer choreography so that both can be tested on CPU with
gloo (tests/test_farm_gloo.py).  No compute happens here.
"""

import os

import torch
import torch.distributed as dist

SEED_BASE = 1000  # std::mt19937(1000 + cascade_or_tile_index)
CASCADE_WAVESCALES = (22.0, 64.0, 176.0, 512.0)

# payload formats: name -> (code of include/datum_ocean_hip.h, torch dtype, elements per grid point)
PAYLOADS = {
    "maps": (0, torch.float32, None),      # per point: texel_bytes / 4 of the module's map layout (capi.map_layout: 6, or 8 in a 32-byte build)
    "xyz32": (1, torch.float32, 3),
    "xyz16": (2, torch.float16, 4),
}


def owned_grids(rank, world, grids_per_rank):
    """Global grid indices rank owns (contiguous block: the gathered buffer is then ordered by global index)."""
    assert 0 <= rank < world
    return list(range(rank * grids_per_rank, (rank + 1) * grids_per_rank))


def grid_seed(global_index):
    return SEED_BASE + global_index


def grid_wavescale(global_index, per_rank):
    return CASCADE_WAVESCALES[(global_index % per_rank) % len(CASCADE_WAVESCALES)]


def map_block_numel(N, grids):
    return grids * 2 * N * N * 4


def payload_numel(N, grids, fmt):
    per = PAYLOADS[fmt][2]
    if per is None:
        from . import capi

        per = capi.map_layout(N)[3] // 4
    return grids * N * N * per


def payload_bytes(N, grids, fmt):
    return payload_numel(N, grids, fmt) * torch.empty(0, dtype=PAYLOADS[fmt][1]).element_size()


def gather_maps(local_maps, world, out=None):
    """All-gather the per-rank blocks (flat tensors of equal size) into one flat tensor ordered by global grid index.
    One collective, on the current stream; with world == 1 it is a copy-free view."""
    if world == 1:
        return local_maps
    if out is None:
        out = torch.empty(world * local_maps.numel(), dtype=local_maps.dtype, device=local_maps.device)
    dist.all_gather_into_tensor(out, local_maps)
    return out


def view_grid(gathered, N, global_index):
    """[2][N][N][4] view of one grid inside a gathered (or local) flat block of LOGICAL maps ([layer][y][x][4], what
    datum_ocean_read_maps returns; the device layout of a bound map buffer is capi.map_layers' business)."""
    n = 2 * N * N * 4
    return gathered[global_index * n:(global_index + 1) * n].view(2, N, N, 4)


def view_displacement(gathered, N, global_index, fmt):
    """(dx, dy, dz) of one grid inside a gathered (or local) xyz32 / xyz16 payload: [N][N][3] view."""
    assert fmt in ("xyz32", "xyz16")
    per = PAYLOADS[fmt][2]
    n = N * N * per
    return gathered[global_index * n:(global_index + 1) * n].view(N, N, per)[..., :3]


def _hip():
    import ctypes

    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64.so" in line:
                return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 is not loaded (no HIP device initialised?)")


def cu_masked_stream(device, mask):
    """A HIP stream whose kernels run only on the compute units whose bit is set in `mask` (hipExtStreamCreateWithCUMask; bit i of
    the mask is CU i / 8 of XCD i % 8 on this part: tools/cumask_probe.py), as a torch stream.  Keeping the collective's copying
    workgroups and the step's workgroups on DIFFERENT compute units takes the collective's bursts out of the in-order memory queues
    the step's workgroups wait in (profiles/r05_gather_overhead.txt)."""
    import ctypes

    hip = _hip()
    # one mask word per 32 compute units of THIS device (256 on an MI355X: 8 words); bits beyond the device are dropped
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    nwords = max(1, (cus + 31) // 32)
    mask &= (1 << cus) - 1
    if mask == 0:
        raise RuntimeError("cu_masked_stream: the mask selects no compute unit of this device")
    words = (ctypes.c_uint32 * nwords)(*[(mask >> (32 * i)) & 0xFFFFFFFF for i in range(nwords)])
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), nwords, words)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    stream = torch.cuda.ExternalStream(handle.value, device=device)
    _MASKED_STREAMS[stream.cuda_stream] = handle.value       # torch does not own an external stream: release_cu_masked_stream destroys it
    return stream


_MASKED_STREAMS = {}


def release_cu_masked_stream(stream):
    """Destroys a stream made by cu_masked_stream once its work is done (torch never destroys an ExternalStream)."""
    handle = _MASKED_STREAMS.pop(getattr(stream, "cuda_stream", None), None)
    if handle:
        stream.synchronize()
        _hip().hipStreamDestroy(ctypes_void_p(handle))


def ctypes_void_p(value):
    import ctypes

    return ctypes.c_void_p(value)


class TileGather:
    """Double-buffered, overlapped all-gather of the ranks' payloads.

    Two slots, each a payload buffer (this rank's contribution) and a gathered buffer (everybody's).  Per batch:

        buf = tg.acquire()                  # slot's payload buffer; the producing stream first waits until the collective
                                            # that last READ this slot has finished
        ... enqueue the pack into buf on the producing (compute) stream ...
        tg.launch()                         # the collective of this slot on the communication stream, ordered behind the
                                            # pack by an event; returns at once, the next batch's kernels overlap it
        ...
        out = tg.result()                   # (any later time) the consuming stream waits for the oldest launched slot
        ... the consumer reads out on ITS stream ...
        tg.release()                        # the consumer's stream is done with it: the slot's next collective waits for this

    On CUDA/HIP the ordering is by events between the producer's stream and an own communication stream; on CPU (gloo)
    the collective is issued with async_op=True and result() waits on its handle.  world == 1: no collective, result()
    returns the payload itself.
    """

    def __init__(self, numel, dtype, device, world, slots=2, standin_peers=0, force_collective=False, standin_workgroups=32, standin_gbps=0.0, comm_cus=0):
        # standin_peers (measurement aid, one GPU only): in place of the collective, the communication stream runs a kernel of
        # `standin_workgroups` workgroups (RCCL's channels are workgroups that copy) that writes the payload `standin_peers`
        # times into the gathered buffer -- the HBM writes of that many peers' tiles arriving -- paced to `standin_gbps` of
        # bus bandwidth (0: as fast as it goes; standin_workgroups == 0: device-to-device copies, which occupy no CU):
        # datum_amd/csrc/farm_standin.hip.  What a busy second stream costs the step, without a node.
        # force_collective (tests): issue the collective in a one-rank process group too
        assert standin_peers == 0 or world == 1
        self.standin = standin_peers
        self.standin_workgroups = standin_workgroups
        self.standin_gbps = standin_gbps
        self.standin_lib = None
        if standin_peers and standin_workgroups:
            import ctypes

            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdatum_farm_standin.so")
            self.standin_lib = ctypes.CDLL(path)      # fails loudly when the library was not built
            self.standin_lib.datum_farm_standin_gather_mode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                                                        ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
            self.standin_mode = int(os.environ.get("DATUM_STANDIN_MODE", "0"))    # tools/gather_overhead.sh: 1 resident only, 2 reads only, 3 writes only
            # tools/gather_overhead.sh: the stand-in as this many launches over equal slices of the payload; launch() issues the first,
            # launch_more() the next one (bench.py spreads them over the batch's steps)
            self.standin_chunks = max(1, int(os.environ.get("DATUM_STANDIN_CHUNKS", "1")))
            self.chunk_state = None
        self.world = world
        self.collective = world > 1 or force_collective
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.payload = [torch.empty(numel, dtype=dtype, device=self.device) for _ in range(slots)]
        parts = world if self.collective else (1 + standin_peers if standin_peers else 0)
        self.gathered = [torch.empty(parts * numel, dtype=dtype, device=self.device) if parts else None for _ in range(slots)]
        self.work = [None] * slots           # gloo: async handles
        self.done = [None] * slots           # cuda: event recorded behind the slot's collective
        self.consumed = [None] * slots       # cuda: event recorded by release() on the consumer's stream
        self.handed = None                   # slot last handed out by result()
        self.head = 0                        # next slot to acquire
        self.acquired = None
        self.pending = []                    # launched, not yet handed out by result()
        if self.cuda:
            # (DATUM_COMM_PRIORITY: tools/gather_overhead.sh -- the communication stream below the compute stream)
            # (DATUM_COMM_CUMASK: the communication stream on its own compute units)
            # comm_cus: the communication stream on that many compute units of its own, the low bits of the device's CU mask
            if comm_cus or "DATUM_COMM_CUMASK" in os.environ:
                self.comm = cu_masked_stream(self.device, (1 << comm_cus) - 1 if comm_cus else int(os.environ["DATUM_COMM_CUMASK"], 16))
            else:
                self.comm = torch.cuda.Stream(self.device, priority=int(os.environ.get("DATUM_COMM_PRIORITY", "0"))) if "DATUM_COMM_PRIORITY" in os.environ else torch.cuda.Stream(self.device)
            self.packed = [torch.cuda.Event() for _ in range(slots)]
            self.timing = [None] * slots     # (start, stop) events around the slot's last collective

    def acquire(self):
        assert self.acquired is None, "launch() the acquired slot first"
        s = self.head
        if self.cuda and self.done[s] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.done[s])   # WAR: the old collective still reads it
        if self.cuda and self.consumed[s] is not None and not (self.collective or self.standin):
            # world == 1, no collective: result() handed out payload[s] itself, so the consumer's release() must order the
            # next PACK of the slot (with a collective it orders the next collective, in launch(): the consumer reads gathered[s])
            torch.cuda.current_stream(self.device).wait_event(self.consumed[s])
            self.consumed[s] = None
        elif self.work[s] is not None:
            self.work[s].wait()
            self.work[s] = None
        if s in self.pending:
            self.pending.remove(s)           # its result was never asked for: overwritten by the new batch
        self.acquired = s
        return self.payload[s]

    def launch(self):
        s = self.acquired
        assert s is not None, "acquire() first"
        self.acquired = None
        self.head = (s + 1) % len(self.payload)
        if self.collective or self.standin:
            if self.cuda:
                self.packed[s].record(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(self.packed[s])
                    if self.consumed[s] is not None:
                        self.comm.wait_event(self.consumed[s])     # WAR: a consumer on another stream may still read gathered[s]
                        self.consumed[s] = None
                    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    start.record(self.comm)
                    if self.collective:
                        dist.all_gather_into_tensor(self.gathered[s], self.payload[s])
                    elif self.standin_lib is not None:
                        self.chunk_state = [s, 0]
                        self._standin_chunk()
                    else:
                        n = self.payload[s].numel()
                        for k in range(1, 1 + self.standin):
                            self.gathered[s][k * n:(k + 1) * n].copy_(self.payload[s], non_blocking=True)
                    stop.record(self.comm)
                    self.done[s] = stop
                    self.timing[s] = (start, stop)
            else:
                self.work[s] = dist.all_gather_into_tensor(self.gathered[s], self.payload[s], async_op=True)
        self.pending.append(s)
        return s

    def _standin_chunk(self):
        s, k = self.chunk_state
        n = self.payload[s].numel()
        es = self.payload[s].element_size()
        per = (n * es // self.standin_chunks) & ~15
        lo = k * per
        nbytes = per if k + 1 < self.standin_chunks else n * es - lo
        rc = self.standin_lib.datum_farm_standin_gather_mode(self.gathered[s][n:].data_ptr() + lo, self.payload[s].data_ptr() + lo, nbytes, self.standin,
                                                             self.standin_workgroups, self.standin_gbps, self.standin_mode, self.comm.cuda_stream)
        assert rc == 0, f"datum_farm_standin_gather: {rc}"
        self.chunk_state[1] = k + 1

    def launch_more(self):
        """stand-in only: the next slice of the last launch()'s transfer, on the communication stream (no-op when all are out)"""
        if self.standin_lib is None or self.chunk_state is None or self.chunk_state[1] >= self.standin_chunks:
            return
        s = self.chunk_state[0]
        with torch.cuda.stream(self.comm):
            self._standin_chunk()
            stop = torch.cuda.Event(enable_timing=True)
            stop.record(self.comm)
            self.done[s] = stop
            self.timing[s] = (self.timing[s][0], stop)

    def result(self):
        """Gathered buffer of the OLDEST launched batch (ordered by global grid index); the current stream waits for it."""
        assert self.pending, "nothing launched"
        s = self.pending.pop(0)
        self.handed = s
        if not self.collective and not self.standin:
            return self.payload[s]
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.done[s])
        elif self.work[s] is not None:
            self.work[s].wait()
            self.work[s] = None
        return self.gathered[s]

    def release(self, slot=None):
        """The consumer (current stream) has finished reading the buffer result() handed out: the slot's next collective
        is ordered behind this point.  Needed when the consumer's stream is not the producing stream (on the producing
        stream the order is already there: acquire / pack / launch of the slot come later on that stream)."""
        s = self.handed if slot is None else slot
        if s is None or not self.cuda:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.consumed[s] = ev

    def last_collective_ms(self, slot):
        """Duration of the slot's last collective on the communication stream (CUDA/HIP only, after it has finished)."""
        if not self.cuda or self.timing[slot] is None:
            return 0.0
        a, b = self.timing[slot]
        b.synchronize()
        return a.elapsed_time(b)

    def drain(self):
        """Host-side wait for everything launched (end of a timed region)."""
        if self.cuda:
            self.comm.synchronize()
        else:
            for s, w in enumerate(self.work):
                if w is not None:
                    w.wait()
                    self.work[s] = None

    def close(self):
        """Waits for everything launched and destroys a CU-masked communication stream (torch does not own an external stream)."""
        self.drain()
        if self.cuda:
            release_cu_masked_stream(self.comm)

"""Multi-rank path on CPU: world_size 2 over gloo.  Checks the tile-farm index arithmetic and the single all-gather that
reassembles the displacement field (datum_amd/farm.py), with the CPU oracle standing in for each rank's GPU (the oracle
is only the data source here; the collective and the layout are what is under test)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def pack_host(logical_maps, fmt):
    """The payload datum_ocean_pack_displacement produces (include/datum_ocean_hip.h), built here from LOGICAL maps
    ([grids][2][N][N][4]) for the CPU tests, where the oracle stands in for a rank's GPU."""
    disp = logical_maps[:, 0, :, :, :3]
    if fmt == "xyz32":
        return disp.contiguous().reshape(-1).to(torch.float32)
    if fmt == "xyz16":
        z = torch.zeros(disp.shape[:-1] + (1,), dtype=disp.dtype)
        return torch.cat([disp, z], -1).to(torch.float16).reshape(-1)
    raise ValueError(fmt)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, per_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from datum_amd import farm
        from oracle import oracle

        mine = farm.owned_grids(rank, world, per_rank)
        local = torch.empty(farm.map_block_numel(N, per_rank), dtype=torch.float32)
        for i, g in enumerate(mine):
            ws = farm.grid_wavescale(g, per_rank)
            _, h0 = oracle.seed(N, farm.grid_seed(g), ws)
            phase = np.zeros((N, N), np.float32)
            m = oracle.displace(h0, phase, ws, 1.35, dt=np.float32(1 / 60))
            farm.view_grid(local, N, i).copy_(torch.from_numpy(m))
        out = farm.gather_maps(local, world)
        # every rank must now hold every grid, ordered by global index
        sums = [float(farm.view_grid(out, N, g)[0, ..., 2].double().abs().sum()) for g in range(world * per_rank)]
        q.put((rank, mine, sums, out.numel()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_farm_and_gather():
    world, N, per_rank = 2, 64, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, sums0, n0), (r1, mine1, sums1, n1) = res
    assert mine0 == [0, 1] and mine1 == [2, 3]
    assert n0 == n1 == world * per_rank * 2 * N * N * 4
    assert np.allclose(sums0, sums1)  # both ranks reassembled the same field
    assert len(set(np.round(sums0, 6))) == 4  # four different grids (own seeds / wavescales)

    # and it is what a single process computes for the same global indices
    from datum_amd import farm
    from oracle import oracle

    for g in range(4):
        ws = farm.grid_wavescale(g, per_rank)
        _, h0 = oracle.seed(N, farm.grid_seed(g), ws)
        m = oracle.displace(h0, np.zeros((N, N), np.float32), ws, 1.35, dt=np.float32(1 / 60))
        assert abs(float(np.abs(m[0, ..., 2].astype(np.float64)).sum()) - sums0[g]) < 1e-6 * max(1.0, sums0[g])


def _pipeline_worker(rank, world, port, N, per_rank, fmt, batches, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from datum_amd import farm
        from oracle import oracle

        mine = farm.owned_grids(rank, world, per_rank)
        states = []
        for g in mine:
            ws = farm.grid_wavescale(g, per_rank)
            _, h0 = oracle.seed(N, farm.grid_seed(g), ws)
            states.append((h0, np.zeros((N, N), np.float32), ws))
        tg = farm.TileGather(farm.payload_numel(N, per_rank, fmt), farm.PAYLOADS[fmt][1], "cpu", world)
        maps = torch.empty(per_rank, 2, N, N, 4)          # the rank's ONE map buffer, overwritten by every batch
        got = []
        for b in range(batches):
            for i, (h0, phase, ws) in enumerate(states):  # "batch b": one more step of every owned grid
                maps[i].copy_(torch.from_numpy(oracle.displace(h0, phase, ws, 1.35, dt=np.float32(1 / 60))))
            tg.acquire().copy_(pack_host(maps, fmt))
            tg.launch()                                   # returns at once; the next batch overwrites `maps` meanwhile
            if b >= 1:
                out = tg.result()                         # batch b - 1
                got.append([farm.view_displacement(out, N, g, fmt).float().clone() for g in range(world * per_rank)])
        out = tg.result()
        got.append([farm.view_displacement(out, N, g, fmt).float().clone() for g in range(world * per_rank)])
        tg.drain()
        q.put((rank, [[float(x.double().abs().sum()) for x in batch] for batch in got], [[x[N // 3, N // 5].tolist() for x in batch] for batch in got]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fmt,world,per_rank", [("xyz32", 2, 2), ("xyz16", 2, 2), ("xyz32", 8, 1)])
def test_pipelined_gather_double_buffer(fmt, world, per_rank):
    # TileGather: the collective of batch k is in flight while batch k + 1 is produced into the other slot and the
    # single map buffer is overwritten; results come out in batch order, every rank sees every grid, and the payload
    # layout ([grid][y][x] (dx, dy, dz) as floats, or (dx, dy, dz, 0) as halves) is what view_displacement reads.
    # (8, 1) = BASELINE.json configs[3]'s indices: eight ranks, one tile each, seeds 1000 ... 1007 -- the rehearsal of the
    # world size no GPU box of the builder's has (the native farm of the C ABI follows the same protocol, slot for slot).
    N, batches = 64, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, N, per_rank, fmt, batches, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    _, sums0, probes0 = res[0]
    assert len(sums0) == batches and all(len(b) == world * per_rank for b in sums0)
    for _, sums, probes in res[1:]:
        assert sums == sums0 and probes == probes0          # every rank reassembled identical fields

    from datum_amd import farm
    from oracle import oracle

    if (world, per_rank) == (8, 1):
        assert [farm.grid_seed(g) for r in range(world) for g in farm.owned_grids(r, world, per_rank)] == list(range(1000, 1008))

    tol = 1e-6 if fmt == "xyz32" else 2e-3
    for g in range(world * per_rank):
        ws = farm.grid_wavescale(g, per_rank)
        _, h0 = oracle.seed(N, farm.grid_seed(g), ws)
        phase = np.zeros((N, N), np.float32)
        for b in range(batches):
            m = oracle.displace(h0, phase, ws, 1.35, dt=np.float32(1 / 60))
            want = m[0, ..., :3].astype(np.float64)
            assert abs(np.abs(want).sum() - sums0[b][g]) < tol * np.abs(want).sum(), (b, g)       # batch b, not a neighbour of it
            assert np.abs(want[N // 3, N // 5] - np.array(probes0[b][g])).max() < tol * max(1.0, np.abs(want).max())


def test_tile_gather_single_rank_passthrough():
    from datum_amd import farm

    tg = farm.TileGather(12, torch.float32, "cpu", 1)
    for k in range(3):
        tg.acquire().fill_(float(k))
        tg.launch()
        assert float(tg.result()[0]) == float(k)
    assert farm.payload_bytes(64, 2, "xyz32") == 2 * 64 * 64 * 12
    assert farm.payload_bytes(64, 2, "xyz16") == 2 * 64 * 64 * 8
    assert farm.payload_bytes(64, 2, "maps") == 2 * 64 * 64 * 24


def test_single_rank_is_a_view():
    from datum_amd import farm

    x = torch.arange(2 * 2 * 8 * 8 * 4, dtype=torch.float32)
    assert farm.gather_maps(x, 1) is x
    assert farm.view_grid(x, 8, 1)[0, 0, 0, 0] == 2 * 8 * 8 * 4
    assert farm.owned_grids(3, 8, 4) == [12, 13, 14, 15]
    assert farm.grid_seed(5) == 1005

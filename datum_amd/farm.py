"""Tile / cascade farm across the GPUs of one node.

Cascades and tiles are independent (OceanParams, N x N grid) problems -- own seed, no halo, each grid is periodic
(data/ocean.map.comp:58) -- so rank r simply owns the global grids [r*C, (r+1)*C) and the displacement step needs no
collective.  The one exchange north_star asks for, "a single RCCL all-gather over xGMI to reassemble the displacement
field", is an all_gather_into_tensor of every rank's [C][2][N][N][4] float map block into a [world*C][2][N][N][4]
buffer; this module holds the index arithmetic so that it can be tested on CPU with gloo.
"""

import torch
import torch.distributed as dist

SEED_BASE = 1000  # std::mt19937(1000 + cascade_or_tile_index)
CASCADE_WAVESCALES = (22.0, 64.0, 176.0, 512.0)


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


def gather_maps(local_maps, world, out=None):
    """All-gather the per-rank map blocks (flat float32 tensors of equal size) into one flat tensor ordered by
    global grid index.  One collective; with world == 1 it is a copy-free view."""
    if world == 1:
        return local_maps
    if out is None:
        out = torch.empty(world * local_maps.numel(), dtype=local_maps.dtype, device=local_maps.device)
    dist.all_gather_into_tensor(out, local_maps)
    return out


def view_grid(gathered, N, global_index):
    """[2][N][N][4] view of one grid inside a gathered (or local) flat map buffer."""
    n = 2 * N * N * 4
    return gathered[global_index * n:(global_index + 1) * n].view(2, N, N, 4)

"""Grid-slab partition and loss-surface exchange for the multi-GPU sweep.

Grid points are independent episodes, so the flat (beta-major) grid is split over the ranks and the only
exchange is one all-gather of the fp64 loss pieces (RCCL when the tensors live on GPUs, gloo on the CPU),
followed by the reference's first-minimum scan on every rank (Calibrate.scala:96,112).  Pieces are padded to a
common length so the gathered parts line up.  Two splits: contiguous slabs, and the interleaved one
(rank r takes points r, r + world, ...) the bench uses -- episodes of a low-beta row cost more than those
of a high-beta row, so contiguous slabs of whole rows leave the ranks unevenly loaded (9 % at 8 ranks on the
64 x 64 grid) while every interleaved share holds the same mix.
"""
from typing import Tuple

import numpy as np


def slab_bounds(grid_points: int, world: int, rank: int) -> Tuple[int, int, int]:
    """(first point, point count, padded slab length) of `rank`."""
    per = (grid_points + world - 1) // world
    g0 = min(grid_points, rank * per)
    cnt = max(0, min(grid_points, g0 + per) - g0)
    return g0, cnt, per


def strided_bounds(grid_points: int, world: int, rank: int) -> Tuple[int, int, int, int]:
    """(first point, stride, point count, padded share length) of `rank` in the interleaved split."""
    per = (grid_points + world - 1) // world
    cnt = max(0, (grid_points - rank + world - 1) // world) if rank < grid_points else 0
    return rank, world, cnt, per


def gather_surface_strided(loss_local, grid_points: int, world: int, force: bool = False):
    """All-gathers the padded interleaved shares and puts them back in flat grid order: the value of grid point
    k * world + r is element k of rank r's share."""
    import torch
    import torch.distributed as dist
    per = loss_local.numel()
    if world > 1 or force:
        full = torch.empty(per * world, dtype=loss_local.dtype, device=loss_local.device)
        dist.all_gather_into_tensor(full, loss_local)
    else:
        full = loss_local
    return full.view(world, per).t().reshape(-1).detach().cpu().numpy()[:grid_points]


def gather_surface(loss_local, grid_points: int, world: int, force: bool = False):
    """All-gathers the padded per-rank slabs (torch tensors of equal length) and returns the surface
    as a host numpy array of length grid_points."""
    import torch
    import torch.distributed as dist
    if world > 1 or force:
        full = torch.empty(loss_local.numel() * world, dtype=loss_local.dtype, device=loss_local.device)
        dist.all_gather_into_tensor(full, loss_local)
    else:
        full = loss_local
    return full.detach().cpu().numpy()[:grid_points]


def first_min(loss: np.ndarray) -> Tuple[int, float]:
    """Strict '<' scan from Double.MaxValue: index and value of the FIRST minimum; (0, MaxValue) when
    nothing is below it (Calibrate.scala:96-97,112)."""
    best, idx = 1.7976931348623157e308, 0
    for g, v in enumerate(np.asarray(loss, dtype=np.float64)):
        if v < best:
            best, idx = float(v), g
    return idx, best

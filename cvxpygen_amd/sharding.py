"""
Multi-GPU use of the batched solver: instances are independent, so a batch is cut into contiguous
shards, one per rank (one process per GPU), the family plan is replicated, and nothing is
exchanged while solving.  The only collective is the FINAL gather of results (RCCL when the
process group uses the `nccl` backend, which is RCCL on ROCm; `gloo` in the CPU tests).

The reference has no counterpart (single process, single thread); this is row
(e) of SURVEY.md section 8.
"""

from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of instances owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_rows(local: np.ndarray, B: int, group=None, device=None) -> np.ndarray:
    """All-gather row blocks of different length (shard_bounds) into the full [B, ...] array on
    every rank.  `device`: torch device for the staging tensors (cuda for nccl/RCCL)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    tail = local.shape[1:]
    maxrows = max(shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world))
    pad = np.zeros((maxrows,) + tail, dtype=local.dtype)
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(B, r, world)
        parts.append(outs[r][:hi - lo].cpu().numpy())
    return np.concatenate(parts, axis=0)


def solve_sharded(solver, theta_var: np.ndarray, group=None, device=None, **kwargs) -> Dict[str, np.ndarray]:
    """Every rank passes the FULL theta_var [B, np_var]; each solves its shard on its own GPU and
    the flat results are gathered on all ranks."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = theta_var.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    res = solver.solve(theta_var=np.ascontiguousarray(theta_var[lo:hi]), B=hi - lo, **kwargs)
    out = {}
    for name, arr in (('prim', res.prim_flat), ('dual', res.dual_flat), ('obj_val', res.obj_val),
                      ('iter', res.iter), ('status', res.status), ('pri_res', res.pri_res),
                      ('dua_res', res.dua_res)):
        out[name] = gather_rows(np.ascontiguousarray(arr), B, group, device)
    return out

# -*- coding: utf-8 -*-
"""Row (cell) sharding across the GPUs of one node: one process per GPU, torch.distributed with
the "nccl" backend (= RCCL over xGMI on ROCm).  The reference has no distributed code; this is
new design.

Cells are independent given V, so rank r owns the contiguous rows [r*n/W, (r+1)*n/W) (remainder to
the last rank) of X, a1, a2, U_hat, log_U_hat (and p_d / D_hat); b1, b2, V_hat, log_V_hat, the
sparsity posteriors and all K-vectors are replicated.  Per sweep the ranks exchange, by sum
all-reduce, only the per-gene accumulators (m x K float32) and the column sums of the new U_hat /
log_U_hat (2 x K float64) -- the U update is row-local, so these partials exist before the V update.
"""
import torch
import torch.distributed as dist

__all__ = ['shard_rows', 'world_size', 'rank', 'all_reduce_sum', 'all_reduce_sum_async', 'sum_int']


def shard_rows(n_total, rank_, world):
    """[r0, r1) of rank `rank_`: floor(n/W) rows each, the remainder goes to the last rank."""
    n_total, rank_, world = int(n_total), int(rank_), int(world)
    if not (0 <= rank_ < world):
        raise ValueError('rank %d outside [0, %d)' % (rank_, world))
    base = n_total // world
    r0 = rank_ * base
    r1 = n_total if rank_ == world - 1 else r0 + base
    return r0, r1


def _active(pg):
    return dist.is_available() and dist.is_initialized() and (pg is not None or dist.get_world_size() > 1)


def world_size(pg=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(pg) if pg is not None else dist.get_world_size()


def rank(pg=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    return dist.get_rank(pg) if pg is not None else dist.get_rank()


def all_reduce_sum(t, pg=None):
    """In-place sum all-reduce (no-op on a single process)."""
    if world_size(pg) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
    return t


def all_reduce_sum_async(t, pg=None):
    """Start an in-place sum all-reduce and return a handle with ``wait()`` (None on a single process).
    On RCCL the collective runs on the communicator's own stream, ordered after the work already queued
    on the current stream, so kernels launched between this call and ``wait()`` overlap with it."""
    if world_size(pg) > 1:
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg, async_op=True)
    return None


def sum_int(v, pg=None, device=None):
    if world_size(pg) == 1:
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
    return int(t.item())

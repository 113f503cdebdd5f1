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

__all__ = ['shard_rows', 'world_size', 'rank', 'all_reduce_sum', 'all_reduce_sum_async', 'sum_int', 'SweepExchange']


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


class SweepExchange:
    """The ONE sum all-reduce of a sweep: a packed float32 buffer

        Z_j (m K) | [Z_log (m K)] | [D_hat^T U_hat: hi (m K) | lo (m K)] | sum_i U_hat, sum_i log U_hat: hi (2 K) | lo (2 K)

    The float32 segments are views the kernels write their per-shard partials into (no copy).  The float64
    quantities travel as two float32 words, hi = float32(x) and lo = float32(x - hi): the pair holds x to
    ~2^-48, and what the float32 all-reduce adds to it is the rounding of the `hi` sums -- the same 1e-7
    relative level as the float32 sums of Z_j next to them (SURVEY 8e compares sharded and single-process
    runs at 1e-6).  On a single process nothing is packed, split or reduced: `get64` hands back the very
    tensor `put64` was given.  The ZI models need one more, small all-reduce per sweep (the column sums of
    p_d, which depend on the post-exchange V_hat, zigap.py:131-132, 158)."""

    def __init__(self, device, pg, f32_shapes, f64_shapes):
        self.pg = pg
        self.world = world_size(pg)
        self.device = torch.device(device)
        self.n_reduces = 0
        self._seg32, self._seg64 = {}, {}
        off = 0
        for name, shape in f32_shapes.items():
            cnt = 1
            for d in shape:
                cnt *= int(d)
            self._seg32[name] = (off, cnt, tuple(int(d) for d in shape))
            off += cnt
        for name, shape in f64_shapes.items():
            cnt = 1
            for d in shape:
                cnt *= int(d)
            self._seg64[name] = (off, cnt, tuple(int(d) for d in shape))
            off += 2 * cnt
        self.numel = off
        self.buf = torch.zeros(max(off, 1), dtype=torch.float32, device=self.device)
        self.f32 = {name: self.buf[o:o + c].view(shape) for name, (o, c, shape) in self._seg32.items()}
        self._local64 = {}
        self._pending = None

    def put64(self, name, t):
        """Stage a float64 tensor of the declared shape (hi / lo split; kept as it is on one process)."""
        if self.world == 1:
            self._local64[name] = t
            return
        o, c, shape = self._seg64[name]
        assert tuple(t.shape) == shape and t.dtype == torch.float64
        flat = t.reshape(-1)
        hi = flat.to(torch.float32)
        self.buf[o:o + c] = hi
        self.buf[o + c:o + 2 * c] = (flat - hi.to(torch.float64)).to(torch.float32)

    def get64(self, name, out=None):
        if self.world == 1:
            t = self._local64[name]
            if out is not None and out is not t:
                out.copy_(t)
                return out
            return t
        o, c, shape = self._seg64[name]
        v = self.buf[o:o + c].to(torch.float64) + self.buf[o + c:o + 2 * c].to(torch.float64)
        v = v.view(shape)
        if out is not None:
            out.copy_(v)
            return out
        return v

    def reduce(self, async_op=False):
        """The collective (no-op on one process).  With async_op the caller must wait()."""
        if self.world == 1:
            return
        self.n_reduces += 1
        if async_op:
            self._pending = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.pg)

    def wait(self):
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

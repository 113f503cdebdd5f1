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
import os

import torch
import torch.distributed as dist

__all__ = ['shard_rows', 'world_size', 'rank', 'sharded', 'all_reduce_sum', 'all_reduce_sum_async', 'sum_int', 'SweepExchange']


def _forced():
    """ORIANA_FORCE_SHARDED=1: a process group of ONE rank takes the sharded code path (every collective of a sweep is
    really issued -- a self all-reduce).  Rehearsal of the RCCL path on a single-GPU box: bench.py
    ORIANA_BENCH_FORCE_PG=1, tests/test_sharded_gpu.py."""
    return os.environ.get('ORIANA_FORCE_SHARDED') == '1'


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


def sharded(pg=None):
    """True when the collectives of a sweep must be issued: more than one rank, or the one-rank rehearsal mode."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return world_size(pg) > 1 or _forced()


def rank(pg=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    return dist.get_rank(pg) if pg is not None else dist.get_rank()


def all_reduce_sum(t, pg=None):
    """In-place sum all-reduce (no-op on a single process)."""
    if sharded(pg):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
    return t


def all_reduce_sum_async(t, pg=None):
    """Start an in-place sum all-reduce and return a handle with ``wait()`` (None on a single process).
    On RCCL the collective runs on the communicator's own stream, ordered after the work already queued
    on the current stream, so kernels launched between this call and ``wait()`` overlap with it."""
    if sharded(pg):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg, async_op=True)
    return None


def sum_int(v, pg=None, device=None):
    if not sharded(pg):
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
    return int(t.item())


class SweepExchange:
    """The exchange of a sweep, issued as ONE step between the column pass and the gene-side update:

        float32 buffer   Z_j (m K) | [Z_log (m K)]                                  -- one sum all-reduce
        float64 buffer   [D_hat^T U_hat (m K)] | sum_i U_hat, sum_i log U_hat (2 K)   -- one (small) sum all-reduce

    The float32 segments are views the kernels write their per-shard partials into (no copy).  The float64 partials
    (the reference forms them in float64: zigap.py:124, gap.py:106, 120-121) are reduced EXACTLY as float64 -- a
    float32 all-reduce of (hi, lo) pairs, as round 2 did, rounds the `hi` sums to 1e-7 and the `lo` words cannot
    recover that.  For pCMF the second collective carries 2 K doubles.  On a single process nothing is packed or
    reduced: `get64` hands back the very tensor `put64` was given.  The ZI models need one more small all-reduce per
    sweep (the column sums of p_d, which depend on the post-exchange V_hat, zigap.py:131-132, 158)."""

    def __init__(self, device, pg, f32_shapes, f64_shapes):
        self.pg = pg
        self.world = world_size(pg)
        self.active = sharded(pg)     # False: nothing is packed or reduced (one process, no rehearsal mode)
        self.device = torch.device(device)
        self.n_reduces = 0            # exchanges (one per sweep)
        self.n_collectives = 0        # all-reduce calls: one per exchange and dtype present
        self._seg32, self._seg64 = {}, {}
        off = 0
        for name, shape in f32_shapes.items():
            cnt = 1
            for d in shape:
                cnt *= int(d)
            self._seg32[name] = (off, cnt, tuple(int(d) for d in shape))
            off += cnt
        self.numel32 = off
        off = 0
        for name, shape in f64_shapes.items():
            cnt = 1
            for d in shape:
                cnt *= int(d)
            self._seg64[name] = (off, cnt, tuple(int(d) for d in shape))
            off += cnt
        self.numel64 = off
        self.numel = self.numel32 + 2 * self.numel64          # in float32 words (what the exchange moves)
        self.buf = torch.zeros(max(self.numel32, 1), dtype=torch.float32, device=self.device)
        self.buf64 = torch.zeros(max(self.numel64, 1), dtype=torch.float64, device=self.device)
        self.f32 = {name: self.buf[o:o + c].view(shape) for name, (o, c, shape) in self._seg32.items()}
        self._local64 = {}
        self._pending = []

    def put64(self, name, t):
        """Stage a float64 tensor of the declared shape (kept as it is on one process)."""
        if not self.active:
            self._local64[name] = t
            return
        o, c, shape = self._seg64[name]
        assert tuple(t.shape) == shape and t.dtype == torch.float64
        self.buf64[o:o + c] = t.reshape(-1)

    def get64(self, name, out=None):
        if not self.active:
            t = self._local64[name]
            if out is not None and out is not t:
                out.copy_(t)
                return out
            return t
        o, c, shape = self._seg64[name]
        v = self.buf64[o:o + c].view(shape)
        if out is not None:
            out.copy_(v)
            return out
        return v.clone()

    def reduce(self, async_op=False):
        """The exchange (no-op on one process): both buffers are handed to the backend back to back (a float64 buffer
        already started by start64() is not sent again).  With async_op the caller must wait()."""
        if not self.active:
            return
        self.n_reduces += 1
        bufs = []
        if self.numel32 and not self._parts32:
            bufs.append(self.buf)
        elif self.numel32:
            # segments were sent early (reduce_rows_async): whatever they did NOT cover is reduced now -- an uncovered
            # range would otherwise enter the gene-side update as this rank's partial sums and the ranks would diverge
            for a, b in self._uncovered():
                bufs.append(self.buf[a:b])
        self._parts32 = False
        self._covered = []
        if self.numel64 and not self._started64:
            bufs.append(self.buf64)
        self._started64 = False
        ev = self._mark() if not async_op else None
        for b in bufs:
            self.n_collectives += 1
            if async_op:
                self._pending.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            else:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.pg)
        if not async_op:
            self.wait()
            if ev is not None:
                self._exposed.append((ev, self._mark()))

    # ---- how long the compute stream stood still at the exchange (bench.py: allreduce_exposed_ms) ------------------------
    timing = False          # set True to bracket every blocking reduce() with events on the current stream
    _exposed = ()

    def _mark(self):
        if not self.timing or self.device.type != 'cuda':
            return None
        if not isinstance(self._exposed, list):
            self._exposed = []
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def exposed_ms(self):
        """Mean time between 'the compute stream reaches the exchange' and 'the last collective of the exchange has completed',
        over the exchanges since timing was switched on (call after a synchronize): what the sweep waits for -- collectives
        started earlier (start64, reduce_rows_async) only count with the part that is still outstanding."""
        pairs = [(a, b) for a, b in self._exposed if a is not None and b is not None]
        self._exposed = []
        if not pairs:
            return None
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)

    _started64 = False
    _parts32 = False
    _covered = ()

    def _uncovered(self):
        """Ranges [a, b) of the float32 buffer (in elements) that no reduce_rows_async call of this exchange has sent."""
        out, pos = [], 0
        for a, b in sorted(self._covered):
            if a > pos:
                out.append((pos, a))
            pos = max(pos, b)
        if pos < self.numel32:
            out.append((pos, self.numel32))
        return out

    def reduce_rows_async(self, name, lo, hi):
        """Start the all-reduce of rows [lo, hi) of the float32 segment `name` NOW, asynchronously (the segment is written
        in an order that makes them final early: engine.zq_gap zj_packed).  The next reduce() then sends no float32 buffer
        and only waits for them -- plus, should the calls of a sweep not cover every row of every float32 segment, one
        all-reduce per uncovered range."""
        if not self.active or hi <= lo:
            return
        o, c, shape = self._seg32[name]
        width = c // shape[0]
        self.n_collectives += 1
        self._pending.append(dist.all_reduce(self.buf[o + lo * width:o + hi * width], op=dist.ReduceOp.SUM, group=self.pg,
                                             async_op=True))
        self._parts32 = True
        if not isinstance(self._covered, list):
            self._covered = []
        self._covered.append((o + lo * width, o + hi * width))

    def start64(self):
        """Start the float64 all-reduce NOW, asynchronously: its partials (the cell-side column sums, D_hat^T U_hat)
        exist before the column pass, which then runs under it; reduce() later sends the float32 buffer only and
        waits for both."""
        if not self.active or not self.numel64:
            return
        self.n_collectives += 1
        self._pending.append(dist.all_reduce(self.buf64, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._started64 = True

    def wait(self):
        for h in self._pending:
            h.wait()
        self._pending = []

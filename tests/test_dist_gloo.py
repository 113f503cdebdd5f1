# -*- coding: utf-8 -*-
"""Row sharding over 2 processes (gloo, CPU): the exchange the GPU models perform -- ONE packed sum
all-reduce per sweep of the per-gene accumulators and of the U_hat / log_U_hat column sums
(oriana_amd/dist.py: SweepExchange) -- reproduces the unsharded sweep.  The per-shard arithmetic is done by the
oracle here (no GPU in this container); the sharding bookkeeping and the collectives are the
product code."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import golden_files, load_golden, state_of, err_colrel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sharded_gap_sweep(rank, world, port, path, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import cavi_oracle as co
        from oriana_amd import dist as odist
        g = load_golden(path)
        s0 = state_of(g, 's0')
        X = g['X']
        n_total, m = X.shape
        K = int(g['meta/k'])
        r0, r1 = odist.shard_rows(n_total, rank, world)
        assert odist.world_size() == world and odist.rank() == rank
        assert odist.sum_int(r1 - r0) == n_total
        Xl = np.ascontiguousarray(X[r0:r1].astype(np.float32))
        lu = np.ascontiguousarray(s0['log_U_hat'][r0:r1]); lv = s0['log_V_hat']
        Zi = np.empty((r1 - r0, K), np.float32); Zj = np.empty((m, K), np.float32)
        co.zq_gap(Zi, Zj, lu, lv, Xl)                                   # local rows only
        # U side is row-local (gap.py:97-102); V_hat column sums are replicated
        a1 = co.clamp(s0['alpha1'][None, :] + Zi)
        a2 = co.clamp(np.broadcast_to(s0['alpha2'] + s0['V_hat'].sum(0), a1.shape).copy())
        U_hat = co.gamma_mean(a1, a2); log_U_hat = co.gamma_meanlog(a1, a2)
        sums = torch.from_numpy(np.stack([U_hat.sum(0), log_U_hat.astype(np.float64).sum(0)]))
        # the exchange: the product's packed buffer, exactly as models/gap.py drives it
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real(*a, **k))[1]
        try:
            xch = odist.SweepExchange('cpu', dist.group.WORLD, {'Zj': (m, K)}, {'sumU': (2, K)})
            xch.f32['Zj'].copy_(torch.from_numpy(Zj))                   # (on the GPU the column pass writes here)
            xch.put64('sumU', sums)
            xch.reduce()
            sums = xch.get64('sumU').clone()
            Zj_t = xch.f32['Zj'].clone()
        finally:
            dist.all_reduce = real
        assert calls == [m * K, 2 * K], calls                           # one exchange: Z_j (float32) + the sums (float64, exact)
        b1 = co.clamp(s0['beta1'][None, :] + Zj_t.numpy())
        b2 = co.clamp(np.broadcast_to(s0['beta2'] + sums[0].numpy(), b1.shape).copy())
        V_hat = co.gamma_mean(b1, b2); log_V_hat = co.gamma_meanlog(b1, b2)
        # M-step from the global sums (gap.py:117-129)
        mean_log_u = (sums[1].numpy() / n_total).astype(np.float32)
        alpha1 = co.clamp(co.inverse_digamma(np.log(s0['alpha2']) + mean_log_u))
        alpha2 = co.clamp(alpha1 / (sums[0].numpy() / n_total))
        gathered = [None] * world
        dist.all_gather_object(gathered, (r0, r1, a1, a2))
        if rank == 0:
            A1 = np.concatenate([t[2] for t in gathered]); A2 = np.concatenate([t[3] for t in gathered])
            assert [t[0] for t in gathered] == sorted(t[0] for t in gathered)
            np.savez(out, a1=A1, a2=A2, b1=b1, b2=b2, alpha1=alpha1, alpha2=alpha2, V_hat=V_hat, log_V_hat=log_V_hat)
    finally:
        dist.destroy_process_group()


def test_two_rank_sweep_matches_unsharded(tmp_path):
    path = golden_files('gap_odd_rand.npz')[0]        # 257 rows: uneven split 128 / 129
    out = str(tmp_path / 'sharded.npz')
    mp.spawn(_sharded_gap_sweep, args=(2, _free_port(), path, out), nprocs=2, join=True)
    got = np.load(out)
    ref = state_of(load_golden(path), 's1')
    # all-reduce changes the summation order only: 1e-6, well inside the 1e-5 budget
    for k in ('a1', 'a2', 'b1', 'b2', 'alpha1', 'alpha2'):
        assert err_colrel(got[k], ref[k]) < 2e-6, k
    assert np.array_equal(got['a2'] == 1e-15, ref['a2'] == 1e-15)


def test_eight_rank_sweep_matches_unsharded(tmp_path):
    """The node's full width: 257 rows over 8 ranks -- 32 rows each, the remainder (33 rows) on the last rank (the
    remainder rule of SURVEY 8e) -- one exchange per sweep, the unsharded sweep reproduced to summation order."""
    from oriana_amd import dist as odist
    shards = [odist.shard_rows(257, r, 8) for r in range(8)]
    assert shards[0] == (0, 32) and shards[-1] == (224, 257)
    assert all(a[1] == b[0] for a, b in zip(shards[:-1], shards[1:]))
    path = golden_files('gap_odd_rand.npz')[0]
    out = str(tmp_path / 'sharded8.npz')
    mp.spawn(_sharded_gap_sweep, args=(8, _free_port(), path, out), nprocs=8, join=True)
    got = np.load(out)
    ref = state_of(load_golden(path), 's1')
    for k in ('a1', 'a2', 'b1', 'b2', 'alpha1', 'alpha2'):
        assert err_colrel(got[k], ref[k]) < 2e-6, k
    assert np.array_equal(got['a2'] == 1e-15, ref['a2'] == 1e-15)


def _exchange_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oriana_amd import dist as odist
        m, K = 37, 5
        rng = np.random.default_rng(100 + rank)
        zj = rng.gamma(2.0, 50.0, size=(m, K)).astype(np.float32)
        zlog = (-rng.gamma(2.0, 500.0, size=(m, K))).astype(np.float32)
        dtu = rng.gamma(2.0, 1e5, size=(m, K))                         # float64, magnitudes of D_hat^T U_hat
        sums = np.stack([rng.gamma(2.0, 1e6, size=K), -rng.gamma(2.0, 1e5, size=K)])
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real(*a, **k))[1]
        xch = odist.SweepExchange('cpu', dist.group.WORLD, {'Zj': (m, K), 'Zlog': (m, K)}, {'DtU': (m, K), 'sumU': (2, K)})
        assert xch.numel32 == 2 * m * K and xch.numel64 == m * K + 2 * K and xch.numel == 2 * m * K + 2 * m * K + 4 * K
        xch.f32['Zj'].copy_(torch.from_numpy(zj)); xch.f32['Zlog'].copy_(torch.from_numpy(zlog))
        xch.put64('DtU', torch.from_numpy(dtu)); xch.put64('sumU', torch.from_numpy(sums))
        xch.reduce()
        dist.all_reduce = real
        assert calls == [xch.numel32, xch.numel64] and xch.n_reduces == 1 and xch.n_collectives == 2
        got = dict(Zj=xch.f32['Zj'].numpy().copy(), Zlog=xch.f32['Zlog'].numpy().copy(),
                   DtU=xch.get64('DtU').numpy().copy(), sumU=xch.get64('sumU').numpy().copy())
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(Zj=zj, Zlog=zlog, DtU=dtu, sumU=sums))
        if rank == 0:
            exact = {k: sum(g[k].astype(np.float64) for g in gathered) for k in got}
            np.savez(out, **{'got_' + k: v for k, v in got.items()}, **{'ref_' + k: v for k, v in exact.items()})
    finally:
        dist.destroy_process_group()


def test_packed_exchange_is_one_step(tmp_path):
    """Z_j | Z_log travel in ONE float32 all-reduce (float32-sum accuracy); D_hat^T U_hat | sum U_hat |
    sum log U_hat -- float64 in the reference -- in one float64 all-reduce issued with it, EXACT to float64 rounding."""
    out = str(tmp_path / 'xch.npz')
    mp.spawn(_exchange_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    for k in ('Zj', 'Zlog'):
        err = np.abs(r['got_' + k] - r['ref_' + k]) / np.abs(r['ref_' + k])
        assert err.max() < 2e-7, (k, err.max())
    for k in ('DtU', 'sumU'):
        err = np.abs(r['got_' + k] - r['ref_' + k]) / np.abs(r['ref_' + k])
        assert err.max() < 1e-15, (k, err.max())


def test_exchange_single_process_is_exact():
    """On one process nothing is packed or rounded: get64 returns the float64 tensor it was given."""
    from oriana_amd import dist as odist
    xch = odist.SweepExchange('cpu', None, {'Zj': (3, 2)}, {'sumU': (2, 2)})
    t = torch.tensor([[1.0 + 2 ** -40, 2.0], [3.0, 4.0]], dtype=torch.float64)
    xch.put64('sumU', t)
    xch.reduce()
    assert xch.n_reduces == 0 and xch.get64('sumU') is t


def _segment_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oriana_amd import dist as odist
        m, K, gd = 37, 4, 11
        rng = np.random.default_rng(200 + rank)
        zj = rng.gamma(2.0, 50.0, size=(m, K)).astype(np.float32)
        sums = np.stack([rng.gamma(2.0, 1e6, size=K), -rng.gamma(2.0, 1e5, size=K)])
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real(*a, **k))[1]
        xch = odist.SweepExchange('cpu', dist.group.WORLD, {'Zj': (m, K)}, {'sumU': (2, K)})
        xch.put64('sumU', torch.from_numpy(sums))
        xch.start64()                                             # before the column pass
        xch.f32['Zj'][gd:].copy_(torch.from_numpy(zj[gd:]))       # the sliced genes' rows are final first ...
        xch.reduce_rows_async('Zj', gd, m)
        xch.f32['Zj'][:gd].copy_(torch.from_numpy(zj[:gd]))       # ... then the dense genes'
        xch.reduce_rows_async('Zj', 0, gd)
        xch.reduce()                                              # nothing left to send: waits
        dist.all_reduce = real
        assert calls == [2 * K, (m - gd) * K, gd * K] and xch.n_reduces == 1 and xch.n_collectives == 3
        got = dict(Zj=xch.f32['Zj'].numpy().copy(), sumU=xch.get64('sumU').numpy().copy())
        # the next sweep without segments falls back to the whole buffer
        xch.put64('sumU', torch.from_numpy(sums))
        xch.reduce()
        assert xch.n_reduces == 2 and xch.n_collectives == 5
        # [r5] a PARTIAL cover (a model with a second float32 segment that only sends part of the first one early): what the
        # early calls did not send is reduced by reduce() itself -- no rank keeps unreduced per-gene sums
        x2 = odist.SweepExchange('cpu', dist.group.WORLD, {'Zj': (m, K), 'Zlog': (m, K)}, {'sumU': (2, K)})
        x2.f32['Zj'].copy_(torch.from_numpy(zj)); x2.f32['Zlog'].copy_(torch.from_numpy(2 * zj))
        x2.put64('sumU', torch.from_numpy(sums))
        x2.reduce_rows_async('Zj', gd, m)
        before = x2.n_collectives
        x2.reduce()
        assert x2.n_collectives == before + 3                     # Zj[:gd], the whole of Zlog, the float64 buffer
        z_all = [None] * world
        dist.all_gather_object(z_all, zj)
        tot = sum(z.astype(np.float64) for z in z_all)
        assert np.allclose(x2.f32['Zj'].numpy(), tot, rtol=3e-7) and np.allclose(x2.f32['Zlog'].numpy(), 2 * tot, rtol=3e-7)
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(Zj=zj, sumU=sums))
        if rank == 0:
            exact = {k: sum(g[k].astype(np.float64) for g in gathered) for k in got}
            np.savez(out, **{'got_' + k: v for k, v in got.items()}, **{'ref_' + k: v for k, v in exact.items()})
    finally:
        dist.destroy_process_group()


def test_exchange_in_segments(tmp_path):
    """The packed per-gene sums travel in two segments, each started as soon as it is final (SweepExchange.reduce_rows_async:
    the sliced genes' rows before the dense gene-side kernel runs); reduce() then only waits.  Sums exact to float32."""
    out = str(tmp_path / 'seg.npz')
    mp.spawn(_segment_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    assert float((np.abs(r['got_Zj'] - r['ref_Zj']) / np.abs(r['ref_Zj'])).max()) < 2e-7
    assert float((np.abs(r['got_sumU'] - r['ref_sumU']) / np.abs(r['ref_sumU'])).max()) < 1e-15

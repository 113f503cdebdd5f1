# -*- coding: utf-8 -*-
"""Row sharding over 2 processes (gloo, CPU): the exchange the GPU models perform -- one sum
all-reduce of the per-gene accumulators and of the U_hat / log_U_hat column sums per sweep
(oriana_amd/dist.py) -- reproduces the unsharded sweep.  The per-shard arithmetic is done by the
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
        Zj_t = torch.from_numpy(Zj)
        odist.all_reduce_sum(Zj_t)                                      # the exchange
        odist.all_reduce_sum(sums)
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

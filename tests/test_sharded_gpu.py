# -*- coding: utf-8 -*-
"""Row-sharded models end to end on the GPU: 2 processes share cuda:0 and exchange through gloo
(RCCL needs one GPU per rank; the collective calls and everything around them are the same code).
The sharded run must reproduce the single-process run (1e-6, summation order only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import golden_files, load_golden, err_colrel

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, name, out, dense_density=None):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import oriana_amd.models as M
        from oriana_amd import dist as odist, engine
        g = load_golden(path)
        X = g['X']; K = int(g['meta/k'])
        r0, r1 = odist.shard_rows(X.shape[0], rank, world)
        dev = torch.device('cuda', 0)
        # every rank packs the genes in the same order: the per-gene counts are all-reduced
        counts = engine.CountTiles.from_dense(X[r0:r1], dev, reduce_fn=lambda t: odist.all_reduce_sum(t),
                                              dense_density=dense_density, n_total=X.shape[0])
        assert (counts.gd >= 32) == bool(dense_density)
        cls = getattr(M, name)
        model = cls(counts, k=K, tau=float(g['meta/tau']), init=(g['s0/a1'][r0:r1], g['s0/b1']), device=dev,
                    process_group=dist.group.WORLD)
        assert model.n_total == X.shape[0]
        model.fit(2)
        metrics = np.array([model.reconstruction_deviance(), model.explained_deviance(), model.frobenius_norm()])
        st = model.state()
        rows = {k: st[k] for k in ('a1', 'a2', 'U_hat') if k in st}
        if 'p_d' in st:
            rows['p_d'] = st['p_d']
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        if rank == 0:
            full = {k: np.concatenate([t[k] for t in gathered]) for k in rows}
            for k in ('b1', 'b2', 'alpha1', 'alpha2', 'beta1', 'beta2', 'pi_d', 'p_s', 'pi_s'):
                if k in st:
                    full[k] = st[k]
            full['metrics'] = metrics
            np.savez(out, **full)
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    # leave without running the interpreter's / c10d's static destructors: with a HIP context and gloo
    # worker threads alive they occasionally race at exit ("terminate called without an active
    # exception"), which mp.spawn would report as a failed rank although the results are written
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


@pytest.mark.parametrize('name,fn', [('GaP', 'gap_odd_rand.npz'), ('ZIGaP', 'zigap_odd_rand.npz'),
                                     ('SparseGaP', 'sparsegap_odd_rand.npz'), ('SparseZIGaP', 'sparsezigap_c1_rand.npz')])
def test_two_ranks_match_one(tmp_path, name, fn):
    import oriana_amd.models as M
    path = golden_files(fn)[0]
    out = str(tmp_path / 'sharded.npz')
    mp.spawn(_worker, args=(2, _free_port(), path, name, out), nprocs=2, join=True)
    got = np.load(out)
    g = load_golden(path)
    single = getattr(M, name)(g['X'], k=int(g['meta/k']), tau=float(g['meta/tau']), init=(g['s0/a1'], g['s0/b1']))
    single.fit(2)
    ref = single.state()
    # Sharding changes the summation order only (float atomics already make it run-dependent).  The
    # Gamma parameters move by ~1e-7 per sweep; the Bernoulli posteriors amplify that through the
    # sigmoid (helpers.exact_twin), so they get an absolute 1e-3 here -- this test is about
    # the sharding logic, the conditioning is covered by tests/test_models_gpu.py.
    for k in got.files:
        if k == 'metrics':
            continue
        tol = 1e-3 if k in ('p_d', 'p_s', 'pi_d', 'pi_s') else 2e-5
        assert err_colrel(got[k], ref[k]) < tol, k
    # the deviances / Frobenius norm are sums over the row shards, all-reduced
    ref_m = np.array([single.reconstruction_deviance(), single.explained_deviance(), single.frobenius_norm()])
    np.testing.assert_allclose(got['metrics'], ref_m, rtol=1e-4)


@pytest.mark.parametrize('name,fn', [('ZIGaP', 'zigap_odd_rand.npz'), ('SparseGaP', 'sparsegap_odd_rand.npz'),
                                     ('SparseZIGaP', 'sparsezigap_c1_rand.npz')])
def test_two_ranks_hybrid_layout_zi_and_sparse(tmp_path, name, fn):
    """The ZI / sparse models row-sharded over two ranks ON THE HYBRID LAYOUT (every gene expressed in >= 10 % of all cells
    dense, the same gene order and dense set on both ranks from the all-reduced counts): the single-process run on the
    sliced layout is reproduced to summation order."""
    import oriana_amd.models as M
    path = golden_files(fn)[0]
    out = str(tmp_path / 'sharded_hybrid.npz')
    mp.spawn(_worker, args=(2, _free_port(), path, name, out, 0.1), nprocs=2, join=True)
    got = np.load(out)
    g = load_golden(path)
    single = getattr(M, name)(g['X'], k=int(g['meta/k']), tau=float(g['meta/tau']), init=(g['s0/a1'], g['s0/b1']))
    assert single.counts.gd == 0
    single.fit(2)
    ref = single.state()
    for k in got.files:
        if k == 'metrics':
            continue
        tol = 1e-3 if k in ('p_d', 'p_s', 'pi_d', 'pi_s') else 2e-5
        assert err_colrel(got[k], ref[k]) < tol, k
    ref_m = np.array([single.reconstruction_deviance(), single.explained_deviance(), single.frobenius_norm()])
    np.testing.assert_allclose(got['metrics'], ref_m, rtol=1e-4)


def _nmf_worker(rank, world, port, X, K, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oriana_amd import dist as odist, engine
        from oriana_amd.models.deviceinit import device_nmf
        r0, r1 = odist.shard_rows(X.shape[0], rank, world)
        ct = engine.CountTiles.from_dense(X[r0:r1], torch.device('cuda', 0), reduce_fn=lambda t: odist.all_reduce_sum(t))
        W, H = device_nmf(ct, K, n_iter=15, tol=0.0, seed=9, pg=dist.group.WORLD)
        gathered = [None] * world
        dist.all_gather_object(gathered, W.cpu().numpy())
        if rank == 0:
            np.savez(out, W=np.concatenate(gathered), H=H.cpu().numpy())
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)


def test_device_nmf_sharded_matches_single(tmp_path):
    """The device NMF start is keyed by global rows: two row shards give the single-process factors
    (up to the summation order of the all-reduced gene-side sums)."""
    from oriana_amd import engine
    from oriana_amd.models.deviceinit import device_nmf
    rng = np.random.default_rng(8)
    X = rng.poisson(rng.gamma(0.5, 3.0, size=(9000, 300))).astype(np.float32)     # > 2 blocks of 4096 rows
    out = str(tmp_path / 'nmf.npz')
    mp.spawn(_nmf_worker, args=(2, _free_port(), X, 5, out), nprocs=2, join=True)
    got = np.load(out)
    W, H = device_nmf(engine.CountTiles.from_dense(X, 'cuda'), 5, n_iter=15, tol=0.0, seed=9)
    assert err_colrel(got['W'], W.cpu().numpy()) < 1e-4
    assert err_colrel(got['H'], H.cpu().numpy()) < 1e-4


def _unpacked_worker(rank, world, port, X, K, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import oriana_amd.models as M
        from oriana_amd import dist as odist
        r0, r1 = odist.shard_rows(X.shape[0], rank, world)
        # the shard is handed over UNPACKED (NumPy): the model packs it and must agree with the other rank on the
        # internal gene order (base.py packed without a reduce_fn)
        model = M.GaP(X[r0:r1], k=K, init='nmf', device=torch.device('cuda', 0), process_group=dist.group.WORLD, seed=4)
        cp = model.counts.col_perm.cpu().numpy()
        model.fit(2)
        gathered = [None] * world
        dist.all_gather_object(gathered, (cp, model.b1.asarray(), model.a1.asarray()))
        if rank == 0:
            assert np.array_equal(gathered[0][0], gathered[1][0]), 'ranks disagree on the gene order'
            assert np.array_equal(gathered[0][1], gathered[1][1]), 'replicated b1 diverged between the ranks'
            np.savez(out, b1=gathered[0][1], a1=np.concatenate([g[2] for g in gathered]))
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)


def test_unpacked_shards_with_device_nmf(tmp_path):
    """init='nmf' on row shards given as plain arrays: every rank packs with the all-reduced gene counts, the
    replicated gene side stays identical on the ranks and matches the single-process model."""
    import oriana_amd.models as M
    rng = np.random.default_rng(12)
    dens = rng.beta(1.0, 4.0, size=330)
    X = (rng.poisson(3.0, size=(700, 330)) * (rng.random((700, 330)) < dens)).astype(np.float32)
    out = str(tmp_path / 'unpacked.npz')
    mp.spawn(_unpacked_worker, args=(2, _free_port(), X, 4, out), nprocs=2, join=True)
    got = np.load(out)
    single = M.GaP(X, k=4, init='nmf', seed=4)
    single.fit(2)
    assert err_colrel(got['b1'], single.b1.asarray()) < 1e-4
    assert err_colrel(got['a1'], single.a1.asarray()) < 1e-4


def _hybrid_worker(rank, world, port, X, K, a1, b1, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import oriana_amd.models as M
        from oriana_amd import dist as odist
        r0, r1 = odist.shard_rows(X.shape[0], rank, world)
        model = M.GaP(X[r0:r1], k=K, init=(a1[r0:r1], b1), device=torch.device('cuda', 0), process_group=dist.group.WORLD,
                      dense_density=0.25, n_total=X.shape[0])
        cp = model.counts.col_perm.cpu().numpy()
        gd = model.counts.gd
        n_coll0 = model._xch.n_collectives
        model.fit(2)
        per_sweep = (model._xch.n_collectives - n_coll0) / 2
        dev = np.array([model.reconstruction_deviance(), model.frobenius_norm()])
        gathered = [None] * world
        dist.all_gather_object(gathered, (cp, gd, model.b1.asarray(), model.a1.asarray(), (r0, r1), per_sweep))
        if rank == 0:
            for g in gathered[1:]:
                assert np.array_equal(gathered[0][0], g[0]), 'ranks disagree on the gene order'
                assert gathered[0][1] == g[1], 'ranks disagree on the dense gene set'
                # (the replicated gene side is recomputed on every rank; its float64 column sums are added with atomics,
                #  so the copies agree to float64 rounding, not bit for bit)
                np.testing.assert_allclose(g[2], gathered[0][2], rtol=1e-12, err_msg='replicated b1 diverged between the ranks')
            np.savez(out, b1=gathered[0][2], a1=np.concatenate([g[3] for g in gathered]), gd=gd,
                     shards=np.array([g[4] for g in gathered]), per_sweep=per_sweep, dev=dev)
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)


def test_four_ranks_hybrid_layout_uneven_split(tmp_path):
    """Four row shards of 5003 cells (1250, 1250, 1250 and the remainder 1253) with the HYBRID layout: every rank picks
    the same dense genes and gene order from the all-reduced counts, one exchange per sweep (the float64 partials and the
    two segments of the packed float32 per-gene sums), and the single-process run is reproduced to summation order.  (A GPU box admits six processes on its card:
    four ranks + this one.)"""
    import oriana_amd.models as M
    rng = np.random.default_rng(21)
    n, m, K = 5003, 400, 100
    dens = np.clip(rng.beta(1.0, 4.0, size=m), 0.01, 1.0)
    X = ((rng.poisson(30.0, size=(n, m)) + 1) * (rng.random((n, m)) < dens)).astype(np.float32)
    a1 = rng.gamma(1.0, size=(n, K)); b1 = rng.gamma(1.0, size=(m, K))
    out = str(tmp_path / 'hybrid.npz')
    mp.spawn(_hybrid_worker, args=(4, _free_port(), X, K, a1, b1, out), nprocs=4, join=True)
    got = np.load(out)
    assert int(got['gd']) >= 32
    assert got['shards'].tolist() == [[0, 1250], [1250, 2500], [2500, 3750], [3750, 5003]]
    # one exchange per sweep, in three collectives: the float64 partials (started before the column pass), the sliced genes'
    # segment of the packed float32 per-gene sums (started before the dense gene-side kernel), the dense genes' segment
    assert float(got['per_sweep']) == 3.0
    single = M.GaP(X, k=K, init=(a1, b1), dense_density=0.25)
    assert single.counts.gd == int(got['gd'])
    single.fit(2)
    assert err_colrel(got['b1'], single.b1.asarray()) < 2e-6
    assert err_colrel(got['a1'], single.a1.asarray()) < 2e-6
    ref = np.array([single.reconstruction_deviance(), single.frobenius_norm()])
    np.testing.assert_allclose(got['dev'], ref, rtol=1e-6)
    # and the sliced layout gives the same sweep
    pure = M.GaP(X, k=K, init=(a1, b1), dense_density=0)
    assert pure.counts.gd == 0
    pure.fit(2)
    assert err_colrel(single.b1.asarray(), pure.b1.asarray()) < 5e-6
    np.testing.assert_allclose(ref, [pure.reconstruction_deviance(), pure.frobenius_norm()], rtol=1e-6)


def test_bench_self_launch_two_ranks():
    """`python3 bench.py --gpus 2` from a bare interpreter (no torchrun): the parent starts two fresh ranks before
    touching the GPU; with ORIANA_BENCH_ONE_GPU=1 both use cuda:0 and gloo stands in for RCCL.  One JSON line, one
    packed collective per sweep."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ORIANA_BENCH_ONE_GPU='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--workload', 'c4_eighth'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    # ... and it is the LAST thing on stdout (RCCL's version banner leaves the C stdio buffer before it, not at exit)
    assert r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-600:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['value'] > 0
    assert d['config']['collectives_per_sweep'] == 3       # one exchange: float64 rate partials + the two segments of the packed per-gene sums
    assert len(d['per_rank_ms']['ranks']) == 2 and d['allreduce_ms'] > 0 and d['allreduce_exposed_ms'] >= 0
    assert d['exchange_bytes'] == (30000 * 100 + 4 * 100) * 4
    assert 'K=100' in d['metric'] and '125k' in d['metric']


def test_bench_under_torch_distributed_run():
    """The driver's own multi-GPU command line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- with two ranks on the one GPU of a test box (ORIANA_BENCH_ONE_GPU=1:
    both use cuda:0, gloo stands in for RCCL): rank 0 prints the one JSON line, and it is the last line of the job's stdout."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ORIANA_BENCH_ONE_GPU='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--workload', 'c2']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-600:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['value'] > 0 and d['scaling'] == 'strong'
    assert len(d['per_rank_ms']['ranks']) == 2


def test_rccl_single_rank_rehearsal():
    """The RCCL path itself, on the one GPU a test box has: `bench.py --gpus 1` with ORIANA_BENCH_FORCE_PG=1 initialises the
    nccl backend (= RCCL) with device_id= exactly as a multi-GPU launch does (bench.py main), and ORIANA_FORCE_SHARDED=1 makes
    the sweep take the sharded code path with a process group of ONE rank -- the counts' all-reduce at packing, the
    asynchronous float64 all-reduce started before the column pass, the float32 all-reduce of Z_j after it, wait(): every
    collective is really issued (a self all-reduce).  Same sweeps, same `check` as the plain single-process run."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'ORIANA_FORCE_SHARDED', 'ORIANA_BENCH_FORCE_PG'):
        base.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '1', '--workload', 'c2',
           '--no-cpu']
    out = {}
    for tag, extra in (('plain', {}), ('rccl', {'ORIANA_BENCH_FORCE_PG': '1', 'MASTER_PORT': str(_free_port())})):
        r = subprocess.run(cmd, env=dict(base, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, r.stdout[-2000:]
        assert r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-600:]
        out[tag] = json.loads(lines[0])
    d = out['rccl']
    assert d['exchange_rehearsal']['backend'] == 'nccl' and d['exchange_rehearsal']['ranks'] == 1
    # every sweep (1 warm-up + 4 timed + the un-instrumented loop of a launch-bound workload) is one exchange of two
    # collectives (float64 partials started before the column pass, float32 per-gene sums after it)
    assert d['exchange_rehearsal']['exchanges'] >= 5 and d['exchange_rehearsal']['exposed_ms'] >= 0
    assert d['exchange_rehearsal']['collectives'] == 2 * d['exchange_rehearsal']['exchanges']      # (c2: sliced layout, one segment)
    assert d['config']['collectives_per_sweep'] == 2
    assert 'exchange_rehearsal' not in out['plain']
    assert abs(d['check'] / out['plain']['check'] - 1.0) < 1e-6

# -*- coding: utf-8 -*-
"""Host logic that needs no GPU: shape algebra (bit-exact), Parameter, seeded init replay,
row sharding bookkeeping."""
import os

import numpy as np
import pytest

from helpers import golden_files, load_golden


def test_dimensions_known_answers():
    """The doctest examples of the reference (oriana/dims.py:91-101)."""
    from oriana_amd import Dimensions
    dims = Dimensions({'n': 10, 'm': 5, 'p': 5, 'k': 3, 'l': 4})
    assert repr(dims('m,k ~ d,s')) == 'Dimension mapping (5, 3) <-> (3, 5, 1)'
    assert repr(dims('n,k ~ s,d')) == 'Dimension mapping (10, 3) <-> (10, 3, 1)'
    assert repr(dims('m,k ~ s,d')) == 'Dimension mapping (5, 3) <-> (5, 3, 1)'
    assert repr(dims('n,m,k ~ d,d,d')) == 'Dimension mapping (10, 5, 3) <-> (1, 150, 1)'
    assert repr(dims('n,k,l,l ~ s,d,c,c')) == 'Dimension mapping (10, 3, 4, 4) <-> (10, 3, 16)'


def test_dimensions_reshape_roundtrip_and_transpose():
    """'n,m,k ~ d,s,d' (reference test/test.py:60-79): tile over m, then a non-trivial permute."""
    from oriana_amd import Dimensions, IncompatibleShapeException
    dims = Dimensions({'n': 2, 'm': 3, 'k': 4})
    rel = dims('n,m,k ~ d,s,d')
    assert rel.shape == (2, 3, 4) and rel.canonical_shape == (3, 8, 1)
    canon = np.arange(24).reshape(3, 8, 1)
    buf = rel.reshape_func(canon)
    assert buf.shape == (2, 3, 4)
    # sample axis (m) is axis 0 of the canonical array
    for j in range(3):
        assert np.array_equal(buf[:, j, :].reshape(-1), canon[j, :, 0])
    assert np.array_equal(rel.inv_reshape_func(buf), canon)
    assert dims('n,k ~ d,d').is_identity() and not rel.is_identity()
    with pytest.raises(IncompatibleShapeException):
        dims('n,k ~ d')
    import torch
    tb = rel.reshape_func(torch.from_numpy(canon))
    assert np.array_equal(tb.numpy(), buf)
    dims['q'] = 7
    assert dims['q'] == 7


def test_parameter_surface():
    """oriana/parameters.py:8-32 surface on a torch buffer (CPU tensor here)."""
    import torch
    from oriana_amd import Parameter
    p = Parameter([[0.02, 0.34], [0.62, 0.79]], device='cpu')
    assert p.shape == (2, 2) and p.asarray().dtype == np.float64
    assert np.array_equal(p[:], np.asarray([[0.02, 0.34], [0.62, 0.79]]))
    p[0, 1] = 5.0
    p[1] = np.asarray([1.0, 2.0])
    assert np.array_equal(p.asarray(), np.asarray([[0.02, 5.0], [1.0, 2.0]]))
    assert p[0, 1] == 5.0
    p.buffer = np.ones((3,))
    assert p.shape == (3,) and isinstance(p.buffer, torch.Tensor) and p.buffer.dtype == torch.float64
    mask = np.asarray([True, False, True])
    p[mask] = 0.5
    assert np.array_equal(p.asarray(), np.asarray([0.5, 1.0, 0.5]))


@pytest.mark.parametrize('path', golden_files(), ids=os.path.basename)
def test_seeded_init_replay(path):
    """np.random.seed(s) + the replayed constructor randomness = the reference's initial a1 / b1."""
    from oriana_amd.models.hostinit import reference_initial_shapes
    g = load_golden(path)
    np.random.seed(int(g['meta/seed']) + 1)
    a1, b1, nmf = reference_initial_shapes(str(g['meta/name']), g['X'], int(g['meta/k']), bool(g['meta/use_factors']))
    assert np.array_equal(np.maximum(1e-15, a1), g['s0/a1'])
    assert np.array_equal(np.maximum(1e-15, b1), g['s0/b1'])
    assert np.array_equal(nmf[0], g['nmf/U'])


def test_shard_rows_bookkeeping():
    """Contiguous blocks, remainder to the last rank, exact cover -- bit-exact."""
    from oriana_amd.dist import shard_rows
    for n, w in ((1000000, 8), (10, 3), (7, 8), (0, 2), (257, 2), (500000, 4)):
        spans = [shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        assert all(b - a == n // w for a, b in spans[:-1])
    assert shard_rows(1000000, 7, 8) == (875000, 1000000)
    with pytest.raises(ValueError):
        shard_rows(10, 3, 3)


def test_count_matrix_surface(tmp_path):
    """oriana_amd.singlecell.CountMatrix keeps the reference surface (cmatrix.py:12-115) and adds
    SciPy-sparse input."""
    import numpy as np
    import pandas as pd
    import scipy.sparse as sp
    from oriana_amd.singlecell import CountMatrix
    from oriana_amd.exceptions import DatatypeException
    X = np.arange(12).reshape(4, 3) % 4
    c = CountMatrix(X)
    assert c.shape == (4, 3) and np.array_equal(c.as_array(), X)
    assert np.array_equal(c.T.as_array(), X.T)
    assert sp.isspmatrix_csc(c.as_sparse_matrix()) and sp.isspmatrix_csr(c.as_sparse_matrix('csr'))
    assert np.array_equal(c.as_sparse_matrix().toarray(), X)
    df = pd.DataFrame(X, index=['a', 'b', 'c', 'd'], columns=['g0', 'g1', 'g2'])
    path = str(tmp_path / 'x.csv')
    df.to_csv(path)
    d = CountMatrix.from_csv(path)
    assert list(d.col_names) == ['g0', 'g1', 'g2'] and list(d.row_names) == ['a', 'b', 'c', 'd']
    assert np.array_equal(d.as_array(), X)
    assert np.array_equal(d['g1'].values, X[:, 1])
    sub = d.filter_rows(['b', 'd'], inplace=False)
    assert sub.shape == (2, 3) and d.shape == (4, 3)
    d.filter_rows(['a'])
    assert d.shape == (1, 3)
    s = CountMatrix(sp.coo_matrix(X))
    assert s.is_sparse and s.shape == (4, 3) and np.array_equal(s.as_array(), X)
    assert np.array_equal(s.T.as_array(), X.T)
    with pytest.raises(DatatypeException):
        CountMatrix([[1, 2], [3, 4]])


def test_hybrid_gene_order_rules():
    """CountTiles.dense_order: dense genes first (density >= threshold, every count an integer below 65535), by
    decreasing non-zero count, cut to a multiple of 32; the rest in decreasing order; a permutation in all cases."""
    import torch
    from oriana_amd.engine import CountTiles
    g = torch.Generator().manual_seed(3)
    m, n = 200, 1000
    nnz = torch.randint(0, n + 1, (m,), generator=g)
    nnz[:5] = 0
    bad = torch.zeros(m, dtype=torch.int64)
    bad[torch.argsort(nnz, descending=True)[:3]] = 1                       # the three densest genes hold a non-integer count
    order, gd = CountTiles.dense_order(nnz, n, bad, 0.5)
    assert sorted(order.tolist()) == list(range(m))
    cand = ((nnz >= 0.5 * n) & (bad == 0)).sum().item()
    assert gd == (cand // 32) * 32 and gd > 0
    head, rest = order[:gd], order[gd:]
    assert (nnz[head] >= 0.5 * n).all() and (bad[head] == 0).all()
    assert (nnz[head][:-1] >= nnz[head][1:]).all() and (nnz[rest][:-1] >= nnz[rest][1:]).all()
    # every candidate that was cut off sorts before any sparser gene, the disqualified ones stay in the sliced part
    assert set(torch.nonzero(bad).flatten().tolist()) <= set(rest.tolist())
    # fewer than 32 candidates: no dense block, plain decreasing order
    order0, gd0 = CountTiles.dense_order(nnz, n, bad, 0.999)
    assert gd0 == 0 and (nnz[order0][:-1] >= nnz[order0][1:]).all()
    # empty genes never qualify, whatever the threshold
    order1, gd1 = CountTiles.dense_order(torch.zeros(64, dtype=torch.int64), n, torch.zeros(64, dtype=torch.int64), 0.0)
    assert gd1 == 0 and sorted(order1.tolist()) == list(range(64))
    # min_share (the ZI / sparse models' 'auto'): the dense block only when its genes hold that share of the non-zeros
    share = float(nnz[head].sum()) / float(nnz.sum())
    o2, g2 = CountTiles.dense_order(nnz, n, bad, 0.5, min_share=share - 0.01)
    assert g2 == gd and torch.equal(o2, order)
    o3, g3 = CountTiles.dense_order(nnz, n, bad, 0.5, min_share=share + 0.01)
    assert g3 == 0 and torch.equal(o3, order0)


def test_hybrid_auto_threshold(monkeypatch):
    """'auto': the measured break-even for a factor count the dense kernels exist for and >= 2e8 entries; the
    ORIANA_DENSE_DENSITY switch overrides the value or turns the layout off (DESIGN_HISTORY.md section 10)."""
    from oriana_amd import engine
    monkeypatch.delenv('ORIANA_DENSE_DENSITY', raising=False)
    assert engine.auto_dense_density(1_000_000, 30_000, 100) == engine.DENSE_DENSITY_DEFAULT
    assert engine.auto_dense_density(10_000, 2_000, 20) is None            # configs[1]: launch-bound, stays sliced
    assert engine.auto_dense_density(1_000_000, 30_000, 112) is None       # no dense kernel above Kp = 100
    assert engine.dense_supported(5) and engine.dense_supported(100) and not engine.dense_supported(101)
    monkeypatch.setenv('ORIANA_DENSE_DENSITY', '0.3')
    assert engine.auto_dense_density(1_000_000, 30_000, 100) == 0.3
    monkeypatch.setenv('ORIANA_DENSE_DENSITY', 'off')
    assert engine.auto_dense_density(1_000_000, 30_000, 100) is None


def test_bench_self_launch_reports_every_rank():
    """`python bench.py --gpus 2` without a launcher starts its ranks itself; when one fails, the parent stops the others
    and says, per rank, its exit code and the last line it wrote (here: no GPU in the build container, so both fail at
    torch.cuda.set_device)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a host without a GPU (the ranks must fail)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--workload', 'c2', '--no-cpu'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    tail = [l for l in r.stderr.splitlines() if l.startswith('[bench] rank ')]
    assert len(tail) == 2 and 'rank 0 rc=' in tail[0] and 'rank 1 rc=' in tail[1], r.stderr[-1500:]
    assert any('first to fail' in l for l in tail)


def _plan(nrb, ncb, K, cost=None):
    import ctypes
    from oriana_amd import _lib
    cm = _lib.OrianaCounts()
    cm.n, cm.m, cm.nrb, cm.ncb = nrb * 256, ncb * 256, nrb, ncb
    sp = _lib.OrianaRowSplit()
    c = None
    if cost is not None:
        c = np.ascontiguousarray(cost, dtype=np.float64)
    rc = _lib.load().oriana_row_pass_plan(ctypes.byref(cm), K, c.ctypes.data if c is not None else None, ctypes.byref(sp))
    assert rc == 0
    return sp.nfull, sp.parts, list(sp.edge[:sp.parts + 1])


def test_row_pass_plan_rounds_of_the_chip(monkeypatch):
    """oriana_row_pass_plan is host arithmetic (no GPU): the two-lane kernels (one work-group per CU) keep the row blocks of
    the full rounds of 256 whole and cut those of the last round into gene ranges of equal cost; short matrices split every
    row block; the other kernels and full last rounds do not split."""
    # the headline: 3907 row blocks = 15 rounds + 67 row blocks, in three ranges (201 work-groups: one third of a round)
    nfull, parts, edges = _plan(3907, 118, 100)
    assert (nfull, parts, edges) == (3840, 3, [0, 39, 79, 118])
    # equal COST: decreasing tile costs move the cut points towards the front
    cost = np.linspace(10.0, 1.0, 118)
    nfull, parts, edges = _plan(3907, 118, 100, cost)
    assert (nfull, parts) == (3840, 3) and edges[0] == 0 and edges[3] == 118 and edges[1] < 39 and edges[2] < 79
    c = np.add.reduceat(cost, edges[:-1])
    assert c.max() / c.min() < 1.15
    # configs[2]: 391 row blocks = 256 whole + 135 in five ranges; configs[4]: 1954 = 1792 + 162 in three
    assert _plan(391, 79, 50)[:2] == (256, 5)
    assert _plan(1954, 98, 64)[:2] == (1792, 3)
    # a last round that is nearly full, or exactly full: nothing to gain
    assert _plan(489, 118, 100)[:2] == (489, 1)
    assert _plan(512, 118, 100)[:2] == (512, 1)
    # kernels with several work-groups per CU (K = 20: one lane per row) keep whole row blocks from 256 on
    assert _plan(391, 79, 20)[:2] == (391, 1)
    # short matrices: every row block split (round 3's rule), evenly cut when there are more than 8 ranges
    nfull, parts, edges = _plan(2, 118, 100)
    assert nfull == 0 and parts == 118 and edges[0] == -1
    assert _plan(40, 8, 20)[:2] == (0, 8)
    # one gene tile: nothing to split
    assert _plan(40, 1, 100)[:2] == (40, 1)


def test_bench_secondaries_cannot_cost_the_headline(monkeypatch):
    """the secondary workloads of the default bench line run in child processes under a wall-clock budget; a
    child that fails (here: no GPU in the build container) or a spent budget becomes an {'error': ...} entry, never an
    exception in the process that holds the measured headline."""
    import importlib.util
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a host without a GPU (the child must fail)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv('ORIANA_BENCH_SECONDARY_BUDGET_S', '120')
    res = bench.run_secondaries(('c3_zi',))
    assert len(res) == 1 and res[0]['workload'] == 'c3_zi' and res[0]['error'].startswith('rc=')
    monkeypatch.setenv('ORIANA_BENCH_SECONDARY_BUDGET_S', '5')
    res = bench.run_secondaries(('c3_zi', 'c5_sparse'))
    assert [r['workload'] for r in res] == ['c3_zi', 'c5_sparse'] and all('skipped' in r['error'] for r in res)

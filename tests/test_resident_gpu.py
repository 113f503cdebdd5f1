# -*- coding: utf-8 -*-
"""[r5] The resident handle of the C ABI through ctypes: the four loop nests (gap.py:67-80, zigap.py:79-95,
sparse_gap.py:81-97, sparse_zigap.py:100-116) on a count matrix packed once by oriana_counts_create_dense_f32 /
oriana_counts_create_csr, against the C oracle; the plans the handle forms against those of engine.CountTiles (same
C planning functions, same layout).  GPU only."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import err_colrel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from oriana_amd import _lib
    assert torch.cuda.is_available()
    return _lib.load()


def _data(seed, n, m, K, z=0.25):
    rng = np.random.default_rng(seed)
    dens = rng.beta(1.0, 1.0 / z - 1.0, size=m)
    X = (rng.poisson(3.0, size=(n, m)) * (rng.random((n, m)) < dens[None, :])).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32)
    lv = rng.normal(size=(m, K)).astype(np.float32)
    return rng, X, lu, lv


def _create(lib, X, K, dd=0.0):
    from oriana_amd._lib import ptr, stream_ptr
    h = ctypes.c_void_p(None)
    Xd = torch.from_numpy(X).cuda()
    rc = lib.oriana_counts_create_dense_f32(ctypes.addressof(h), ptr(Xd), X.shape[0], X.shape[1], X.shape[1], K, float(dd), stream_ptr())
    assert rc == 0 and h.value
    return h


def _info(lib, h):
    info = (ctypes.c_int64 * 13)()
    assert lib.oriana_counts_info(h, info, 13) == 0
    return list(info)


@pytest.mark.parametrize('n,m,K,dd', [(300, 270, 5, 0.0), (1000, 600, 20, 0.0), (1000, 600, 20, 0.15), (520, 257, 50, 0.0),
                                      (777, 300, 64, 0.2), (1500, 420, 100, 0.0), (1500, 420, 100, 0.15), (40, 33, 7, 0.0),
                                      (600, 64, 36, 1e-6)])
def test_zq_gap_resident_matches_oracle(lib, n, m, K, dd):
    from oracle import cavi_oracle as co
    from oriana_amd._lib import ptr, stream_ptr
    rng, X, lu, lv = _data(n + K, n, m, K)
    h = _create(lib, X, K, dd)
    info = _info(lib, h)
    assert info[:3] == [n, m, K] and info[4] == int(np.count_nonzero(X))
    assert (info[5] > 0) == (dd > 0) and info[5] % 32 == 0
    if dd == 1e-6:
        assert info[5] == (int((X != 0).any(0).sum()) // 32) * 32          # every expressed gene dense
    Zi, Zj = torch.full((n, K), 7.0, device='cuda'), torch.full((m, K), 7.0, device='cuda')      # (callee zero-fills)
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    for rep in range(3):
        if rep:
            lu = (lu + rng.normal(size=lu.shape) * 0.3).astype(np.float32)
        lud, lvd = torch.from_numpy(lu).cuda(), torch.from_numpy(lv).cuda()
        assert lib.oriana_zq_gap_resident(h, ptr(Zi), ptr(Zj), ptr(lud), ptr(lvd), stream_ptr()) == 0
        torch.cuda.synchronize()
        co.zq_gap(rZi, rZj, lu, lv, X)
        assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5, rep
    assert lib.oriana_counts_destroy(h) == 0


def test_resident_plans_equal_the_python_hosts(lib):
    """engine.CountTiles (the Python host) and the handle call the same planning functions on the same tables: same gene
    order, same dense set, same slots, same number of column work items, same row split."""
    from oriana_amd import engine
    n, m, K = 2600, 900, 100
    _, X, lu, lv = _data(3, n, m, K, z=0.3)
    for dd in (0.0, 0.2):
        h = _create(lib, X, K, dd)
        info = _info(lib, h)
        ct = engine.CountTiles.from_dense(X, 'cuda', dense_density=dd or None)
        ws = engine.ZWorkspace(ct, K)
        w = ct.col_work_for(K)
        assert info[5] == ct.gd and info[6] == ct.rslots and info[7] == ct.cslots and info[4] == ct.nnz
        assert info[9] == (0 if w is None else w.shape[0])
        assert info[10] == ws.row_split.parts and info[11] == ws.row_split.nfull
        assert info[12] == int(lib.oriana_device_cus())
        lib.oriana_counts_destroy(h)


def test_resident_from_csr_equals_dense(lib):
    import scipy.sparse as sp
    from oriana_amd._lib import ptr, stream_ptr
    n, m, K = 900, 410, 20
    _, X, lu, lv = _data(11, n, m, K)
    A = sp.csr_matrix(X)
    indptr = np.ascontiguousarray(A.indptr.astype(np.int64))
    indices = np.ascontiguousarray(A.indices.astype(np.int32))
    data = np.ascontiguousarray(A.data.astype(np.float32))
    lud, lvd = torch.from_numpy(lu).cuda(), torch.from_numpy(lv).cuda()
    outs = []
    for make in ('dense', 'csr'):
        if make == 'dense':
            h = _create(lib, X, K, 0.1)
        else:
            h = ctypes.c_void_p(None)
            assert lib.oriana_counts_create_csr(ctypes.addressof(h), indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, n, m, K,
                                                0.1, stream_ptr()) == 0
        info = _info(lib, h)
        Zi, Zj = torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda')
        assert lib.oriana_zq_gap_resident(h, ptr(Zi), ptr(Zj), ptr(lud), ptr(lvd), stream_ptr()) == 0
        torch.cuda.synchronize()
        outs.append((info[4:8], Zi.cpu().numpy(), Zj.cpu().numpy()))
        lib.oriana_counts_destroy(h)
    assert outs[0][0] == outs[1][0]
    # (same layout, same plans; the sums differ by the order of the float atomics only: the dense row kernel of a short
    #  matrix splits its gene tiles over several work-groups, the column pass flushes with atomics)
    assert err_colrel(outs[1][1], outs[0][1]) < 1e-6 and err_colrel(outs[1][2], outs[0][2]) < 1e-6


@pytest.mark.parametrize('dd', [0.0, 0.2], ids=['sliced', 'hybrid'])
@pytest.mark.parametrize('K', [20, 50, 64, 100])
def test_twins_resident_match_oracle(lib, K, dd):
    """[r6] zigap.py:79-95 (both index conventions of :94; with and without its third output), sparse_gap.py:81-97,
    sparse_zigap.py:100-116 under oriana_counts_declare_unit_dropout -- D_hat == 1 at the non-zero counts, as every D_hat of the
    reference's own models is -- on a sliced and on a HYBRID handle, through the kernels the model classes use (K = 20: four
    lanes per row; 50, 64: the two-lane k64 kernels, fused two-image row pass + dual column pass for the sparse nests; 100:
    k100 + the four-kernel sparse form), against the C oracle with the same D_hat; a dead gene; repeated calls on one handle."""
    from oracle import cavi_oracle as co
    from oriana_amd._lib import ptr, stream_ptr
    n, m = 900, 420
    rng, X, lu, lv = _data(200 + K, n, m, K, z=0.3)
    D = rng.random((n, m)).astype(np.float32)
    D[X != 0] = 1.0                                                        # zigap.py:135 + bernoulli.py:45
    St = (rng.random((m, K)) < 0.7).astype(np.float32)
    dead = int(np.argsort((X != 0).sum(0))[m // 2])
    St[dead] = 0.0                                                         # a gene with no active factor (sliced part)
    St[int(np.argmax((X != 0).sum(0)))] = 0.0                              # ... and the densest gene (dense block of the hybrid handle)
    Sh = rng.random((m, K)).astype(np.float32)
    h = _create(lib, X, K, dd)
    assert (_info(lib, h)[5] > 0) == (dd > 0)
    assert lib.oriana_counts_declare_unit_dropout(h, 1) == 0
    d = lambda a: torch.from_numpy(a).cuda()
    lud, lvd, Dd, Std, Shd = d(lu), d(lv), d(D), d(St), d(Sh)
    Zi, Zj, Zl = torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')
    rZi, rZj, rZl = np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)
    st = stream_ptr()
    for quirk in (1, 0):
        co.zq_zigap(rZi, rZj, rZl, lu, lv, D, X, quirk=bool(quirk))
        for with_log in (True, False):
            Zl.fill_(7.0)
            assert lib.oriana_zq_zigap_resident(h, ptr(Zi), ptr(Zj), ptr(Zl) if with_log else None, ptr(lud), ptr(lvd), ptr(Dd), quirk, st) == 0
            torch.cuda.synchronize()
            assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5, (quirk, with_log)
            if with_log:
                assert err_colrel(Zl.cpu().numpy(), rZl) < 2e-5, quirk
            else:
                assert bool((Zl == 7.0).all())                            # NULL: the log sums are skipped, nothing written
    for rep in range(2):                                                   # (twice: the handle's scratch of the first call is reused)
        assert lib.oriana_zq_sparse_gap_resident(h, ptr(Zi), ptr(Zj), ptr(Zl), ptr(lud), ptr(lvd), ptr(Std), ptr(Shd), st) == 0
        torch.cuda.synchronize()
        co.zq_sparse_gap(rZi, rZj, rZl, lu, lv, St, Sh, X)
        assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5
        assert err_colrel(Zl.cpu().numpy(), rZl) < 2e-5 and not Zj[dead].any()
    assert lib.oriana_zq_sparse_zigap_resident(h, ptr(Zi), ptr(Zj), ptr(Zl), ptr(lud), ptr(lvd), ptr(Std), ptr(Shd), ptr(Dd), st) == 0
    torch.cuda.synchronize()
    co.zq_sparse_zigap(rZi, rZj, rZl, lu, lv, St, Sh, D, X)
    assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5
    assert err_colrel(Zl.cpu().numpy(), rZl) < 2e-5
    # the pCMF nest still runs on the same handle afterwards
    assert lib.oriana_zq_gap_resident(h, ptr(Zi), ptr(Zj), ptr(lud), ptr(lvd), st) == 0
    torch.cuda.synchronize()
    co.zq_gap(rZi, rZj, lu, lv, X)
    assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5
    # without the declaration: a general D_hat on the sliced handle (below), ORIANA_EUNIT on the hybrid one
    assert lib.oriana_counts_declare_unit_dropout(h, 0) == 0
    rc = lib.oriana_zq_zigap_resident(h, ptr(Zi), ptr(Zj), ptr(Zl), ptr(lud), ptr(lvd), ptr(Dd), 1, st)
    assert rc == (-4 if dd > 0 else 0)
    lib.oriana_counts_destroy(h)


@pytest.mark.parametrize('K', [7, 20, 50, 64, 100])
def test_twins_resident_general_weights(lib, K):
    """zigap.py:79-95 (both index conventions of :94), sparse_gap.py:81-97, sparse_zigap.py:100-116 with a general D_hat
    (gathered at the stored entries on every call) and a dead gene."""
    from oracle import cavi_oracle as co
    from oriana_amd._lib import ptr, stream_ptr
    n, m = 640, 300
    rng, X, lu, lv = _data(100 + K, n, m, K)
    D = rng.random((n, m)).astype(np.float32)
    St = (rng.random((m, K)) < 0.7).astype(np.float32)
    St[5] = 0.0                                                            # a gene with no active factor
    Sh = rng.random((m, K)).astype(np.float32)
    h = _create(lib, X, K, 0.0)
    d = lambda a: torch.from_numpy(a).cuda()
    lud, lvd, Dd, Std, Shd = d(lu), d(lv), d(D), d(St), d(Sh)
    Zi, Zj, Zl = torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')
    rZi, rZj, rZl = np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)
    st = stream_ptr()
    for quirk in (1, 0):
        assert lib.oriana_zq_zigap_resident(h, ptr(Zi), ptr(Zj), ptr(Zl), ptr(lud), ptr(lvd), ptr(Dd), quirk, st) == 0
        torch.cuda.synchronize()
        co.zq_zigap(rZi, rZj, rZl, lu, lv, D, X, quirk=bool(quirk))
        assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5, quirk
        assert err_colrel(Zl.cpu().numpy(), rZl) < 2e-5, quirk
    assert lib.oriana_zq_sparse_gap_resident(h, ptr(Zi), ptr(Zj), ptr(Zl), ptr(lud), ptr(lvd), ptr(Std), ptr(Shd), st) == 0
    torch.cuda.synchronize()
    co.zq_sparse_gap(rZi, rZj, rZl, lu, lv, St, Sh, X)
    assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5
    assert err_colrel(Zl.cpu().numpy(), rZl) < 2e-5 and not Zj[5].any()
    assert lib.oriana_zq_sparse_zigap_resident(h, ptr(Zi), ptr(Zj), ptr(Zl), ptr(lud), ptr(lvd), ptr(Std), ptr(Shd), ptr(Dd), st) == 0
    torch.cuda.synchronize()
    co.zq_sparse_zigap(rZi, rZj, rZl, lu, lv, St, Sh, D, X)
    assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5
    assert err_colrel(Zl.cpu().numpy(), rZl) < 2e-5
    # and the pCMF nest still runs on the same handle afterwards
    assert lib.oriana_zq_gap_resident(h, ptr(Zi), ptr(Zj), ptr(lud), ptr(lvd), st) == 0
    torch.cuda.synchronize()
    co.zq_gap(rZi, rZj, lu, lv, X)
    assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5
    lib.oriana_counts_destroy(h)


def test_resident_argument_errors(lib):
    from oriana_amd._lib import ptr, stream_ptr
    h = ctypes.c_void_p(None)
    X = torch.zeros(8, 8, device='cuda')
    assert lib.oriana_counts_create_dense_f32(ctypes.addressof(h), None, 8, 8, 8, 5, 0.0, stream_ptr()) == -1
    assert lib.oriana_counts_create_dense_f32(ctypes.addressof(h), ptr(X), 8, 8, 4, 5, 0.0, stream_ptr()) == -1      # ldx < m
    assert lib.oriana_counts_create_dense_f32(ctypes.addressof(h), ptr(X), 8, 8, 8, 1000, 0.0, stream_ptr()) == -2   # K range
    assert lib.oriana_counts_create_dense_f32(ctypes.addressof(h), ptr(X), 8, 8, 8, 5, 0.0, stream_ptr()) == 0       # all zeros: fine
    Zi, Zj = torch.ones(8, 5, device='cuda'), torch.ones(8, 5, device='cuda')
    l = torch.zeros(8, 5, device='cuda')
    assert lib.oriana_zq_gap_resident(h, ptr(Zi), ptr(Zj), ptr(l), ptr(l), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert not Zi.any() and not Zj.any()
    assert lib.oriana_zq_gap_resident(h, None, ptr(Zj), ptr(l), ptr(l), stream_ptr()) == -1
    assert lib.oriana_counts_destroy(h) == 0 and lib.oriana_counts_destroy(None) == 0
    # a hybrid handle serves the ZI nests under the unit declaration only (the dense-gene kernels carry no per-entry weights)
    rng = np.random.default_rng(0)
    Xh = (rng.random((300, 64)) < 0.6).astype(np.float32)
    hh = _create(lib, Xh, 20, 0.3)
    assert _info(lib, hh)[5] >= 32
    Zl = torch.empty(64, 20, device='cuda')
    D = torch.ones(300, 64, device='cuda')
    Zi2, Zj2, l1, l2 = torch.empty(300, 20, device='cuda'), torch.empty(64, 20, device='cuda'), torch.zeros(300, 20, device='cuda'), torch.zeros(64, 20, device='cuda')
    assert lib.oriana_zq_zigap_resident(hh, ptr(Zi2), ptr(Zj2), ptr(Zl), ptr(l1), ptr(l2), ptr(D), 0, stream_ptr()) == -4
    assert lib.oriana_counts_declare_unit_dropout(hh, 1) == 0 and lib.oriana_counts_declare_unit_dropout(None, 1) == -1
    assert lib.oriana_zq_zigap_resident(hh, ptr(Zi2), ptr(Zj2), ptr(Zl), ptr(l1), ptr(l2), ptr(D), 0, stream_ptr()) == 0
    assert lib.oriana_zq_zigap_resident(hh, ptr(Zi2), ptr(Zj2), ptr(Zl), ptr(l1), ptr(l2), None, 0, stream_ptr()) == -1
    torch.cuda.synchronize()
    lib.oriana_counts_destroy(hh)


def test_resident_edge_shapes_and_ineligible_genes(lib):
    """Ragged shapes (one cell, one gene, n and m off every tile boundary), K = 1, genes whose counts do not fit the uint16 block
    (>= 65535, non-integer: they stay on the sliced layout of a hybrid handle), CSR with duplicate entries and empty rows."""
    from oracle import cavi_oracle as co
    from oriana_amd._lib import ptr, stream_ptr
    import scipy.sparse as sp
    rng = np.random.default_rng(42)
    # (70,001 cells x 40 genes: more rows in one packing chunk than a grid's y dimension holds)
    for n, m, K, dd in ((1, 1, 1, 0.0), (1, 300, 3, 0.0), (257, 1, 2, 0.0), (33, 65, 5, 0.5), (513, 259, 20, 0.3), (70001, 40, 4, 0.0)):
        X = (rng.poisson(2.0, size=(n, m)) * (rng.random((n, m)) < 0.6)).astype(np.float32)
        if m > 40:
            X[:, 3] = 70000.0                       # too large for the uint16 block
            X[0, 7] = 2.5                            # not an integer
        lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
        h = _create(lib, X, K, dd)
        info = _info(lib, h)
        assert info[4] == int(np.count_nonzero(X))
        if dd > 0 and m > 40:
            assert info[5] % 32 == 0 and info[5] <= int(((X != 0).mean(0) >= dd).sum())
        Zi, Zj = torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda')
        assert lib.oriana_zq_gap_resident(h, ptr(Zi), ptr(Zj), ptr(torch.from_numpy(lu).cuda()), ptr(torch.from_numpy(lv).cuda()), stream_ptr()) == 0
        torch.cuda.synchronize()
        rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
        co.zq_gap(rZi, rZj, lu, lv, X)
        assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5, (n, m, K, dd)
        lib.oriana_counts_destroy(h)
    # CSR: duplicates add up, empty rows, an out-of-range gene index is an argument error
    n, m, K = 300, 90, 7
    rows = rng.integers(0, n, size=4000); cols = rng.integers(0, m, size=4000)
    rows[rows % 11 == 0] = 5                                      # many duplicates, and rows that stay empty
    vals = rng.integers(1, 6, size=4000).astype(np.float32)
    order = np.argsort(rows, kind='stable')
    rows, cols, vals = rows[order], cols[order], vals[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    indptr = np.cumsum(indptr)
    X = np.zeros((n, m), dtype=np.float32)
    np.add.at(X, (rows, cols), vals)
    assert np.array_equal(sp.csr_matrix((vals, cols, indptr), shape=(n, m)).toarray(), X)
    indices = np.ascontiguousarray(cols.astype(np.int32)); data = np.ascontiguousarray(vals)
    h = ctypes.c_void_p(None)
    assert lib.oriana_counts_create_csr(ctypes.addressof(h), indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, n, m, K, 0.0, stream_ptr()) == 0
    assert _info(lib, h)[4] == int(np.count_nonzero(X))
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    Zi, Zj = torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda')
    assert lib.oriana_zq_gap_resident(h, ptr(Zi), ptr(Zj), ptr(torch.from_numpy(lu).cuda()), ptr(torch.from_numpy(lv).cuda()), stream_ptr()) == 0
    torch.cuda.synchronize()
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    co.zq_gap(rZi, rZj, lu, lv, X)
    assert err_colrel(Zi.cpu().numpy(), rZi) < 1e-5 and err_colrel(Zj.cpu().numpy(), rZj) < 1e-5
    lib.oriana_counts_destroy(h)
    bad = indices.copy(); bad[10] = m
    h2 = ctypes.c_void_p(None)
    assert lib.oriana_counts_create_csr(ctypes.addressof(h2), indptr.ctypes.data, bad.ctypes.data, data.ctypes.data, n, m, K, 0.0, stream_ptr()) == -1
    r = int(np.nonzero(np.diff(indptr)[1:] > 0)[0][0]) + 1                  # a non-monotone indptr: row r "ends before it starts"
    badp = indptr.copy(); badp[r] = indptr[r + 1] + 1
    assert lib.oriana_counts_create_csr(ctypes.addressof(h2), badp.ctypes.data, indices.ctypes.data, data.ctypes.data, n, m, K, 0.0, stream_ptr()) == -1

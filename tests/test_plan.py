# -*- coding: utf-8 -*-
"""[r5] The planning of the resident layout as C host code (csrc/resident.hip: oriana_plan_gene_order, oriana_plan_col_work,
oriana_plan_dense_splits, oriana_row_pass_plan_cus) against NumPy restatements of what oriana_amd/engine.py did in Python
through round 4.  Index bookkeeping: bit-exact.  No GPU (the functions take and return host arrays); the reference has no
counterpart (its loop nest walks the dense matrix, gap.py:72-80)."""
import ctypes
import os

import numpy as np
import pytest


@pytest.fixture(scope='module')
def lib():
    from oriana_amd import _lib
    return _lib.load()


# ---- NumPy restatements (round 4's engine.py) ---------------------------------------------------------------------------
def np_gene_order(col_nnz, n_total, bad, density, min_share=0.0):
    order = np.argsort(-col_nnz, kind='stable')
    if not density:
        return order, 0
    ok = (col_nnz.astype(np.float64) >= float(density) * max(int(n_total), 1)) & (bad == 0) & (col_nnz > 0)
    cand = order[ok[order]]
    gd = (len(cand) // 32) * 32
    if gd and min_share > 0.0:
        tot = float(col_nnz.sum())
        if tot <= 0.0 or float(col_nnz[cand[:gd]].sum()) < float(min_share) * tot:
            gd = 0
    if gd == 0:
        return order, 0
    keep = np.ones(len(order), dtype=bool)
    keep[cand[:gd]] = False
    return np.concatenate([cand[:gd], order[keep[order]]]), gd


def np_col_work(nit, nrb, ncb, width, cus=256, target_items=None, rounds=True, sum_price=False):
    nit = nit.reshape(nrb, ncb).astype(np.float64)
    nblk = (ncb + width - 1) // width
    if width > 1:
        pad = np.zeros((nrb, nblk * width - ncb))
        nit = np.concatenate([nit, pad], axis=1).reshape(nrb, nblk, width)
        nit = nit.sum(axis=2) if sum_price else nit.max(axis=2)
    cost = nit * 1.45 + 3.2
    cums = [np.concatenate([[0.0], np.cumsum(cost[:, cb])]) for cb in range(nblk)]
    total = 0.0
    for cum in cums:
        total += cum[-1]
    explicit = target_items
    nt = nrb * ncb
    if target_items is None:
        target_items = min(36 * cus, max(9 * cus, nt // (50 * width)))

    def build(n_items):
        target = max(total / n_items, 1e-9)
        out = []
        for cb in range(nblk):
            cum = cums[cb]
            nb = int(min(nrb, max(1, round(cum[-1] / target))))
            step = cum[-1] / nb
            pts = np.asarray([i * step for i in range(1, nb)])
            edges = np.unique(np.searchsorted(cum, pts, side='left'))
            edges = np.concatenate([[0], edges, [nrb]]).astype(np.int64)
            edges = np.unique(edges)
            for a, e in zip(edges[:-1], edges[1:]):
                if e > a:
                    out.append((cb, int(a), int(e)))
        return out
    items = build(target_items)
    if explicit is None and rounds and len(items) > cus:
        want = (len(items) // cus) * cus
        slack, back = max(1, 24 * cus // 256), max(1, 8 * cus // 256)
        t = target_items
        trial = items
        for it in range(9):
            if want - slack <= len(trial) <= want:
                items = trial
                break
            if it == 8:
                break
            t = max(cus, int(round(t * (want - back) / max(len(trial), 1))))
            trial = build(t)
    items.sort(key=lambda x: (x[1] + x[2], x[0]))
    return np.asarray(items, dtype=np.int32).reshape(-1, 3)


def c_col_work(lib, nit, nrb, ncb, width, cus=256, target_items=0, rounds=1, sum_price=0):
    nit = np.ascontiguousarray(nit, dtype=np.int32)
    cap = int(lib.oriana_plan_col_work_capacity(nrb, ncb, width))
    items = np.empty((max(cap, 1), 3), dtype=np.int32)
    n = ctypes.c_int64(0)
    rc = lib.oriana_plan_col_work(nit.ctypes.data, nrb, ncb, width, cus, target_items, rounds, sum_price, items.ctypes.data, cap,
                                  ctypes.addressof(n))
    assert rc == 0
    return items[:n.value].copy()


# ---- gene order -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('seed', range(6))
def test_gene_order_matches_numpy(lib, seed):
    rng = np.random.default_rng(seed)
    m, n = int(rng.integers(1, 900)), 5000
    col_nnz = rng.integers(0, n + 1, size=m).astype(np.int64)
    col_nnz[rng.random(m) < 0.2] = col_nnz[0]                      # ties: the caller's order decides
    bad = (rng.random(m) < 0.05).astype(np.int64)
    for density, share in ((0.0, 0.0), (0.3, 0.0), (0.6, 0.0), (0.3, 0.5), (0.3, 0.999), (1e-9, 0.0)):
        order = np.empty(m, dtype=np.int32)
        gd = ctypes.c_int64(-1)
        rc = lib.oriana_plan_gene_order(col_nnz.ctypes.data, bad.ctypes.data, m, n, density, share, order.ctypes.data, ctypes.addressof(gd))
        assert rc == 0
        ref, gd_ref = np_gene_order(col_nnz, n, bad, density, share)
        assert gd.value == gd_ref and np.array_equal(order, ref), (density, share)
        assert sorted(order.tolist()) == list(range(m))
    # bad = NULL: every gene may go to the dense block
    order = np.empty(m, dtype=np.int32)
    gd = ctypes.c_int64(-1)
    assert lib.oriana_plan_gene_order(col_nnz.ctypes.data, None, m, n, 0.3, 0.0, order.ctypes.data, ctypes.addressof(gd)) == 0
    ref, gd_ref = np_gene_order(col_nnz, n, np.zeros(m, np.int64), 0.3)
    assert gd.value == gd_ref and np.array_equal(order, ref)


def test_gene_order_argument_errors(lib):
    gd = ctypes.c_int64(0)
    assert lib.oriana_plan_gene_order(None, None, 5, 10, 0.0, 0.0, None, ctypes.addressof(gd)) == -1
    assert lib.oriana_plan_gene_order(None, None, 0, 10, 0.0, 0.0, None, ctypes.addressof(gd)) == 0 and gd.value == 0


# ---- column work list -----------------------------------------------------------------------------------------------------
def _tiles(rng, nrb, ncb):
    """Longest column slices of a matrix whose genes are packed by decreasing density (what the packer leaves)."""
    dens = np.sort(rng.beta(1.0, 9.0, size=ncb))[::-1]
    base = rng.poisson(lam=np.maximum(dens * 40.0, 0.05), size=(nrb, ncb))
    return base.astype(np.int32).reshape(-1)


@pytest.mark.parametrize('nrb,ncb,width', [(3907, 102, 2), (489, 118, 2), (391, 79, 2), (40, 8, 1), (1954, 98, 2), (700, 30, 1),
                                           (3, 1, 2), (1, 5, 1), (260, 7, 2)])
@pytest.mark.parametrize('cus', [256, 128, 304])
def test_col_work_matches_numpy(lib, nrb, ncb, width, cus):
    rng = np.random.default_rng(nrb * 131 + ncb)
    nit = _tiles(rng, nrb, ncb)
    for kw in (dict(), dict(rounds=0), dict(sum_price=1), dict(target_items=700)):
        got = c_col_work(lib, nit, nrb, ncb, width, cus, **kw)
        ref = np_col_work(nit, nrb, ncb, width, cus, target_items=kw.get('target_items') or None, rounds=bool(kw.get('rounds', 1)),
                          sum_price=bool(kw.get('sum_price', 0)))
        assert got.shape == ref.shape and np.array_equal(got, ref), (kw, got.shape, ref.shape)
        # a partition: every (column block, row block) exactly once
        nblk = (ncb + width - 1) // width
        cover = np.zeros((nblk, nrb), dtype=np.int32)
        for cb, a, e in got:
            assert 0 <= a < e <= nrb
            cover[cb, a:e] += 1
        assert (cover == 1).all()


def test_col_work_lands_on_whole_rounds(lib):
    """One 1024-thread group per CU: the item count is re-cut to just below a multiple of the CU count (DESIGN 10 l)."""
    rng = np.random.default_rng(7)
    nit = _tiles(rng, 3907, 102)
    for cus in (128, 256, 304):
        n = len(c_col_work(lib, nit, 3907, 102, 2, cus))
        first = len(c_col_work(lib, nit, 3907, 102, 2, cus, rounds=0))
        assert n % cus == 0 or cus - n % cus <= max(1, 24 * cus // 256), (cus, n)
        assert abs(n - first) < cus


def test_col_work_empty_and_errors(lib):
    n = ctypes.c_int64(5)
    assert lib.oriana_plan_col_work(None, 0, 4, 2, 256, 0, 1, 0, None, 0, ctypes.addressof(n)) == 0 and n.value == 0
    assert lib.oriana_plan_col_work(None, 3, 4, 2, 256, 0, 1, 0, None, 0, ctypes.addressof(n)) == -1
    nit = np.ones(12, dtype=np.int32)
    items = np.empty((1, 3), dtype=np.int32)
    assert lib.oriana_plan_col_work(nit.ctypes.data, 3, 4, 2, 0, 0, 1, 0, items.ctypes.data, 1, ctypes.addressof(n)) == -1     # cus = 0
    assert lib.oriana_plan_col_work(nit.ctypes.data, 3, 4, 2, 256, 700, 1, 0, items.ctypes.data, 1, ctypes.addressof(n)) == -1   # capacity


# ---- row split and dense splits for other CU counts --------------------------------------------------------------------------
def _row_plan(lib, nrb, ncb, K, cus, cost=None):
    from oriana_amd import _lib
    cm = _lib.OrianaCounts()
    cm.n, cm.m, cm.nrb, cm.ncb = nrb * 256, ncb * 256, nrb, ncb
    sp = _lib.OrianaRowSplit()
    c = np.ascontiguousarray(cost, dtype=np.float64) if cost is not None else None
    rc = lib.oriana_row_pass_plan_cus(ctypes.byref(cm), K, c.ctypes.data if c is not None else None, cus, ctypes.byref(sp))
    assert rc == 0
    return sp.nfull, sp.parts, list(sp.edge[:sp.parts + 1])


def test_row_plan_for_128_256_304_compute_units(lib):
    """The rounds of the chip are rounds of ITS compute units (a literal 256 silently inverts the optimisation on a
    partitioned or differently binned part)."""
    # the headline shape: 3907 row blocks
    assert _row_plan(lib, 3907, 118, 100, 256) == (3840, 3, [0, 39, 79, 118])
    nfull, parts, _ = _row_plan(lib, 3907, 118, 100, 128)          # 30 rounds of 128 + 67: 67 x 2 = 134 -> 2 rounds / 2 ...
    assert nfull == 3840 and parts >= 2
    # 304 CUs: 12 rounds + 259 row blocks; in 7 ranges the last round is 1813 short groups = 5.96 rounds of a seventh
    assert _row_plan(lib, 3907, 118, 100, 304)[:2] == (12 * 304, 7)
    # a full last round on the one part is a partly filled one on the other
    assert _row_plan(lib, 512, 118, 100, 256)[:2] == (512, 1)
    assert _row_plan(lib, 512, 118, 100, 304)[0] == 304
    assert _row_plan(lib, 608, 118, 100, 304)[:2] == (608, 1)
    # short matrices: two work-groups per CU at most
    assert _row_plan(lib, 40, 8, 20, 256)[:2] == (0, 8)
    assert _row_plan(lib, 40, 8, 20, 128)[:2] == (0, 4)            # 6 wanted, whole tiles per range: 8 tiles in 4 ranges of 2
    assert lib.oriana_row_pass_plan_cus(None, 100, None, 0, None) == -1


def test_dense_splits(lib):
    def plan(n, gd, cus):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        assert lib.oriana_plan_dense_splits(n, gd, cus, ctypes.addressof(a), ctypes.addressof(b)) == 0
        return a.value, b.value
    # round 4's Python rule at 256 CUs: ceil(512 / row blocks) gene ranges, ceil(1024 / groups of 8 gene tiles) cell ranges (x 8)
    for n, gd in ((1_000_000, 4064), (125_000, 4064), (10_000, 320), (300, 32), (5003, 96)):
        nblk, ngt = max((n + 255) // 256, 1), gd // 32
        groups = (ngt + 7) // 8
        ref = (max(1, min(ngt, -(-512 // nblk))), max(1, min((n + 31) // 32, (-(-1024 // groups) + 7) // 8 * 8)))
        assert plan(n, gd, 256) == ref
    assert plan(1_000_000, 4064, 128)[0] == 1 and plan(10_000, 320, 304)[0] == min(10, -(-608 // 40))
    a = ctypes.c_int64(0)
    assert lib.oriana_plan_dense_splits(10, 33, 256, ctypes.addressof(a), ctypes.addressof(a)) == -1


def test_device_cus_without_a_device(lib, monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a device is visible')
    assert int(lib.oriana_device_cus()) == 256

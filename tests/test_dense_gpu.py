# -*- coding: utf-8 -*-
"""The hybrid layout: the densest genes evaluated on the bf16 matrix cores (csrc/dense_pass.hip), the rest on the
sliced non-zero layout -- parity of GaP.compute_Z_q_expectations (gap.py:67-80) with the CPU oracle and the
reference's golden kernel outputs, through the C ABI.  GPU only."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_files, load_golden, err_colrel

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope='module')
def eng():
    from oriana_amd import engine
    assert torch.cuda.is_available()
    return engine


def _counts(rng, n, m, dens, maxv=3000):
    """Genes of very different density; counts up to a few thousand (the generator's range)."""
    X = rng.poisson(40.0, size=(n, m)).astype(np.int64) + 1
    X[rng.random((n, m)) < 0.02] = maxv
    X *= (rng.random((n, m)) < dens[None, :])
    return X


def _run(eng, X, lu, lv, density, K):
    from oracle import cavi_oracle as co
    n, m = X.shape
    ct = eng.CountTiles.from_dense(X, 'cuda', dense_density=density)
    ws = eng.ZWorkspace(ct, K)
    Zi = torch.empty(n, K, dtype=torch.float32, device='cuda')
    Zj = torch.empty(m, K, dtype=torch.float32, device='cuda')
    eng.zq_gap(ws, Zi, Zj, torch.from_numpy(lu).cuda(), torch.from_numpy(lv).cuda())
    torch.cuda.synchronize()
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    co.zq_gap(rZi, rZj, lu, lv, np.ascontiguousarray(X.astype(np.float32)))
    return Zi.cpu().numpy(), Zj.cpu().numpy(), rZi, rZj, ct, ws


@pytest.mark.parametrize('n,m,dtype', [(257, 131, np.int64), (600, 300, np.float32), (64, 33, np.int32), (1, 40, np.float64)])
def test_hybrid_pack_roundtrip(eng, n, m, dtype):
    rng = np.random.default_rng(7 * n + m)
    dens = rng.beta(1.0, 3.0, size=m)
    X = _counts(rng, n, m, dens).astype(dtype)
    X[:, 3] = 70000 * (X[:, 3] != 0)               # a gene whose counts do not fit 16 bits stays on the sliced side
    ct = eng.CountTiles.from_dense(X, 'cuda', dense_density=0.2)
    assert ct.gd % 32 == 0 and ct.gd + ct.ms == m
    assert ct.nnz == int((X != 0).sum())
    assert np.array_equal(ct.to_dense(), X.astype(np.float32))
    cp = ct.col_perm.cpu().numpy()
    assert np.array_equal(np.sort(cp), np.arange(m))
    if ct.gd:
        assert 3 not in cp[:ct.gd]
        cnt = (X != 0).sum(0)
        assert (cnt[cp[:ct.gd]] >= 0.2 * n).all()
        with pytest.raises(Exception):
            ct.c_struct                              # consumers that know nothing of dense genes are refused


@pytest.mark.parametrize('n,m,K', [(257, 131, 100), (300, 200, 96), (512, 64, 100), (100, 70, 20), (90, 40, 5),
                                   (600, 260, 50), (257, 300, 64), (130, 96, 37), (33, 64, 84), (700, 150, 68),
                                   (64, 64, 16), (257, 131, 72)])
def test_zq_gap_hybrid_vs_oracle(eng, n, m, K):
    rng = np.random.default_rng(n + 31 * m + 977 * K)
    dens = np.clip(rng.beta(1.0, 2.0, size=m), 0.01, 1.0)
    dens[:36] = np.linspace(1.0, 0.4, 36)
    X = _counts(rng, n, m, dens)
    lu = (rng.normal(size=(n, K)) * 1.5).astype(np.float32)
    lv = (rng.normal(size=(m, K)) * 1.5 - 1.0).astype(np.float32)
    Zi, Zj, rZi, rZj, ct, ws = _run(eng, X, lu, lv, 0.3, K)
    assert ct.gd >= 32 and (ct.ms > 0 or m % 32 == 0)
    assert err_colrel(Zi, rZi) < RTOL
    assert err_colrel(Zj, rZj) < RTOL
    # conservation: sum_k Z_i[i, k] = sum_j x_ij (every responsibility vector sums to 1)
    assert np.allclose(Zi.sum(1), X.sum(1), rtol=2e-6)
    assert np.allclose(Zj.sum(1), X.sum(0), rtol=2e-5, atol=1e-3)


def test_all_genes_dense(eng):
    rng = np.random.default_rng(5)
    n, m, K = 300, 96, 100
    X = _counts(rng, n, m, np.full(m, 0.7))
    lu = rng.normal(size=(n, K)).astype(np.float32)
    lv = rng.normal(size=(m, K)).astype(np.float32)
    Zi, Zj, rZi, rZj, ct, ws = _run(eng, X, lu, lv, 0.1, K)
    assert ct.gd == m and ct.ms == 0
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL


@pytest.mark.parametrize('path', golden_files('gap_*.npz'), ids=os.path.basename)
def test_zq_gap_golden_hybrid(eng, path):
    """Against the REFERENCE's own kernel output on the post-init state, every gene with >= 10 % expression dense
    (the use_factors=True goldens live on the exact slow path: rejected factor rows, den == 0)."""
    g = load_golden(path)
    Zi, Zj, rZi, rZj, ct, ws = _run(eng, g['X'], g['s0/log_U_hat'], g['s0/log_V_hat'], 0.1, g['s0/log_U_hat'].shape[1])
    assert ct.gd >= 32
    assert err_colrel(Zi, g['kernel/Zi']) < RTOL
    assert err_colrel(Zj, g['kernel/Zj']) < RTOL
    assert err_colrel(Zi, rZi) < RTOL
    assert err_colrel(Zj, rZj) < RTOL


def test_dense_slow_path_rows_and_tiny_denominators(eng):
    """Shifts outside the accepted window (rows replaced by the constant), -inf logs and denominators under the
    threshold: the dense side flags them and the exact slow path reproduces the reference's float32 loop."""
    rng = np.random.default_rng(11)
    n, m, K = 200, 96, 100
    X = _counts(rng, n, m, np.full(m, 0.6))
    lu = rng.normal(size=(n, K)).astype(np.float32)
    lv = rng.normal(size=(m, K)).astype(np.float32)
    lu[5] += 35.0; lu[17] -= 80.0; lv[3] += 30.0; lv[40] = -120.0      # (sums stay below 88: no overflow in the reference)
    lu[30, :] = -np.inf
    lu[31, ::2] = -np.inf
    # disjoint supports: den of (cell 40.., gene 50..) is ~ e^-60
    lu[40:44] = -30.0; lu[40:44, :10] = 0.0
    lv[50:54] = -30.0; lv[50:54, 90:] = 0.0
    Zi, Zj, rZi, rZj, ct, ws = _run(eng, X, lu, lv, 0.1, K)
    assert ct.gd == m
    assert int(ws.dn_flag.sum().item()) > 0
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL


def test_hybrid_equals_sliced_layout(eng):
    """The two layouts of the same matrix agree to float32 rounding (1e-6), far inside the tolerance."""
    rng = np.random.default_rng(3)
    n, m, K = 1500, 700, 100
    dens = np.clip(rng.beta(1.0, 9.0, size=m), 0.005, 1.0)
    X = _counts(rng, n, m, dens)
    lu = torch.from_numpy(rng.normal(size=(n, K)).astype(np.float32)).cuda()
    lv = torch.from_numpy(rng.normal(size=(m, K)).astype(np.float32)).cuda()
    out = []
    for dd in (None, 0.1):
        ct = eng.CountTiles.from_dense(X, 'cuda', dense_density=dd)
        ws = eng.ZWorkspace(ct, K)
        Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda')
        eng.zq_gap(ws, Zi, Zj, lu, lv)
        out.append((Zi.cpu().numpy(), Zj.cpu().numpy(), ct.gd))
    assert out[0][2] == 0 and out[1][2] >= 32
    assert err_colrel(out[1][0], out[0][0]) < 2e-6
    assert err_colrel(out[1][1], out[0][1]) < 2e-6


@pytest.mark.parametrize('n,m,K,quirk', [(257, 131, 7, True), (300, 200, 50, True), (400, 160, 100, True), (300, 200, 64, False)])
def test_zq_zigap_hybrid_vs_oracle(eng, n, m, K, quirk):
    """zigap.py:79-95 as the model runs it (D_hat = 1 at the non-zero counts, the D_hat[i, k] index of zigap.py:94 under
    reference_quirks) through the hybrid layout against the C oracle; also with a cell on the exact slow path."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(n + 13 * m + K)
    dens = np.clip(rng.beta(1.0, 2.0, size=m), 0.01, 1.0)
    dens[:40] = np.linspace(1.0, 0.4, 40)
    X = _counts(rng, n, m, dens).astype(np.float32)
    lu = (rng.normal(size=(n, K)) * 1.5).astype(np.float32)
    lv = (rng.normal(size=(m, K)) * 1.5 - 1.0).astype(np.float32)
    lu[3] -= 80.0
    D = rng.random((n, m)).astype(np.float32)
    D[X != 0] = 1.0                                   # zigap.py:135 + bernoulli.py:45
    ct = eng.CountTiles.from_dense(X, 'cuda', dense_density=0.3)
    assert ct.gd >= 32
    ws = eng.ZWorkspace(ct, K)
    Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda')
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    dq = c(D[:, :K]) if quirk else None
    eng.zq(ws, Zi, Zj, None, c(lu), c(lv), dq=dq)
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    co.zq_zigap(r[0], r[1], r[2], lu, lv, D, X, quirk=quirk)
    assert int(ws.dn_flag.sum().item()) > 0
    assert err_colrel(Zi.cpu().numpy(), r[0]) < RTOL
    assert err_colrel(Zj.cpu().numpy(), r[1]) < RTOL


@pytest.mark.parametrize('n,m,K', [(300, 200, 50), (257, 160, 100), (400, 131, 64), (130, 96, 7)])
def test_zq_sparse_hybrid_vs_oracle(eng, n, m, K):
    """sparse_gap.py:81-97 through the hybrid layout against the C oracle: den against the masked FV image, the S_hat-weighted
    row sums against FV * S_hat (oriana_dense_images2), both per-gene sums and the log sums; a dead gene inside the dense
    block (every factor off: its tiles take the exact slow path and contribute zeros), a cell on the slow path."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(n + 17 * m + K)
    dens = np.clip(rng.beta(1.0, 2.0, size=m), 0.01, 1.0)
    dens[:40] = np.linspace(1.0, 0.4, 40)
    X = _counts(rng, n, m, dens).astype(np.float32)
    lu = (rng.normal(size=(n, K))).astype(np.float32)
    lv = (rng.normal(size=(m, K)) - 0.5).astype(np.float32)
    lu[3] -= 80.0
    ps = rng.random((m, K))
    St = (ps > 0.3).astype(np.float32); Sh = ps.astype(np.float32)
    St[5] = 0                                        # a dead gene among the densest
    ct = eng.CountTiles.from_dense(X, 'cuda', dense_density=0.3)
    assert ct.gd >= 32 and 5 in ct.col_perm.cpu().numpy()[:ct.gd]
    ws = eng.ZWorkspace(ct, K)
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    o = [torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')]
    eng.zq(ws, o[0], o[1], o[2], c(lu), c(lv), S_tilde=c(St), S_hat=c(Sh))
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    co.zq_sparse_gap(r[0], r[1], r[2], lu, lv, St, Sh, X)
    for got, ref in zip(o, r):
        assert err_colrel(got.cpu().numpy(), ref) < RTOL
    assert not o[1].cpu().numpy()[5].any()
    # and the sliced layout of the same matrix agrees to float32 rounding
    ct0 = eng.CountTiles.from_dense(X, 'cuda')
    ws0 = eng.ZWorkspace(ct0, K)
    o0 = [torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')]
    eng.zq(ws0, o0[0], o0[1], o0[2], c(lu), c(lv), S_tilde=c(St), S_hat=c(Sh))
    for a, b in zip(o, o0):
        assert err_colrel(a.cpu().numpy(), b.cpu().numpy()) < 5e-6

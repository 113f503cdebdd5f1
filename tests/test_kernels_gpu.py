# -*- coding: utf-8 -*-
"""Parity of the HIP responsibility pass (through the C ABI) with the CPU oracle.  GPU only."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_files, load_golden, err_colrel, dropout_sweep

pytestmark = pytest.mark.gpu

RTOL = 1e-5   # north_star tolerance; the sums are float32 on both sides


@pytest.fixture(scope='module')
def eng():
    from oriana_amd import engine
    assert torch.cuda.is_available()
    return engine


def _rand_counts(rng, n, m, density, maxv=50):
    X = rng.poisson(3.0, size=(n, m)).astype(np.int64) + 1
    X *= (rng.random((n, m)) < density)
    X[rng.random((n, m)) < 0.01 * density] = maxv * 1000
    return X


@pytest.mark.parametrize('n,m,density,dtype', [(257, 131, 0.3, np.int64), (600, 700, 0.1, np.float32),
                                               (256, 256, 1.0, np.int32), (1, 1, 1.0, np.float64),
                                               (513, 5, 0.5, np.int64), (300, 300, 0.0, np.float32)])
def test_pack_roundtrip(eng, n, m, density, dtype):
    rng = np.random.default_rng(n * 1000 + m)
    X = _rand_counts(rng, n, m, density).astype(dtype)
    ct = eng.CountTiles.from_dense(X, 'cuda', sort_rows=(n % 2 == 1))       # both row orders over the cases
    assert ct.nnz == int((X != 0).sum())
    assert np.array_equal(ct.to_dense(), X.astype(np.float32))
    if ct.row_perm is not None:
        # cells are ordered by decreasing depth inside a packing chunk (a permutation of the rows)
        rp = ct.row_perm.cpu().numpy()
        assert np.array_equal(np.sort(rp), np.arange(n))
        depth = (X != 0).sum(1)[rp]
        assert (np.diff(depth) <= 0).all() or n > 8192
    # column-side structures are consistent with the row-side records
    if ct.nnz:
        h = ct.host_arrays()
        Xp = X.astype(np.float32)
        if ct.col_perm is not None:
            Xp = Xp[:, ct.col_perm.cpu().numpy()]
        if ct.row_perm is not None:                       # packed row r holds cell row_perm[r]
            Xp = Xp[ct.row_perm.cpu().numpy()]
        seen_total = 0
        for t in range(ct.nrb * ct.ncb):
            rb, cb = divmod(t, ct.ncb)
            rs, cs = h['rslice'][t], h['cslice'][t]
            assert h['roff'][t + 1] - h['roff'][t] == rs[16]
            assert h['coff'][t + 1] - h['coff'][t] == cs[16] + 64          # + dummy slots
            assert (rs % 64 == 0).all() and (cs % 64 == 0).all()
            rec = h['rec'][h['roff'][t]:h['roff'][t + 1]]
            rid = h['ridx'][h['coff'][t]:h['coff'][t + 1]]
            slot = np.arange(len(rec))
            sl = np.searchsorted(rs, slot, side='right') - 1
            row = sl * 16 + (((slot - rs[sl]) & 63) >> 2)
            keep = rec['x'] != 0
            seen_total += int(keep.sum())
            cd = rec['cdst'][keep].astype(np.int64)
            assert len(np.unique(cd)) == len(cd) and (cd < cs[16]).all()
            # the column-side slot of an entry belongs to the entry's column and names its row
            csl = np.searchsorted(cs, cd, side='right') - 1
            colc = csl * 16 + (((cd - cs[csl]) & 63) >> 2)
            assert np.array_equal(colc, rec['col'][keep].astype(np.int64))
            assert np.array_equal(rid[cd].astype(np.int64), row[keep])
            gr = rb * 256 + row[keep]; gc = cb * 256 + rec['col'][keep].astype(np.int64)
            assert np.array_equal(Xp[gr, gc], rec['x'][keep])
        assert seen_total == ct.nnz


def _same_zeros(got, ref):
    """Where one side is exactly 0 the other is at most denormal dust (< 1e-30 of the column max):
    the factorised exponentials underflow at slightly different points than exp(lu + lv)."""
    thr = 1e-30 * np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-30)
    assert (np.abs(got)[ref == 0] <= np.broadcast_to(thr, ref.shape)[ref == 0]).all()
    assert (np.abs(ref)[got == 0] <= np.broadcast_to(thr, ref.shape)[got == 0]).all()


def _run_gap(eng, X, lu, lv):
    from oracle import cavi_oracle as co
    n, m = X.shape
    K = lu.shape[1]
    ct = eng.CountTiles.from_dense(X, 'cuda')
    ws = eng.ZWorkspace(ct, K)
    Zi = torch.empty(n, K, dtype=torch.float32, device='cuda')
    Zj = torch.empty(m, K, dtype=torch.float32, device='cuda')
    eng.zq_gap(ws, Zi, Zj, torch.from_numpy(lu).cuda(), torch.from_numpy(lv).cuda())
    torch.cuda.synchronize()
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    co.zq_gap(rZi, rZj, lu, lv, np.ascontiguousarray(X.astype(np.float32)))
    return Zi.cpu().numpy(), Zj.cpu().numpy(), rZi, rZj, ws


@pytest.mark.parametrize('path', golden_files('gap_*.npz'), ids=os.path.basename)
def test_zq_gap_golden(eng, path):
    """Against the REFERENCE's own kernel output (golden) and the oracle, on the post-init state."""
    g = load_golden(path)
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, g['X'], g['s0/log_U_hat'], g['s0/log_V_hat'])
    assert err_colrel(Zi, g['kernel/Zi']) < RTOL
    assert err_colrel(Zj, g['kernel/Zj']) < RTOL
    assert err_colrel(Zi, rZi) < RTOL
    assert err_colrel(Zj, rZj) < RTOL
    # zero patterns (den == 0 guard / exp underflow) must agree, up to float32 denormal dust
    _same_zeros(Zi, g['kernel/Zi'])
    _same_zeros(Zj, g['kernel/Zj'])


@pytest.mark.parametrize('n,m,K,density', [(300, 270, 5, 0.3), (257, 131, 7, 0.5), (512, 300, 20, 0.1),
                                          (400, 520, 50, 0.1), (300, 257, 64, 0.2), (700, 300, 100, 0.1),
                                          (260, 300, 128, 0.1), (300, 260, 200, 0.05), (257, 258, 256, 0.05),
                                          (1, 300, 3, 1.0), (300, 1, 3, 1.0)])
def test_zq_gap_random(eng, n, m, K, density):
    rng = np.random.default_rng(K * 7 + n)
    X = _rand_counts(rng, n, m, density)
    lu = (rng.normal(size=(n, K)) * 2.0).astype(np.float32)
    lv = (rng.normal(size=(m, K)) * 2.0 + 1.0).astype(np.float32)
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, X, lu, lv)
    assert int(ws.tile_flag.sum().item()) == 0          # everything on the fast path
    assert err_colrel(Zi, rZi) < RTOL
    assert err_colrel(Zj, rZj) < RTOL
    # size-independent property: responsibilities sum to the counts
    np.testing.assert_allclose(Zi.sum(1), X.sum(1), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(Zj.sum(1), X.sum(0), rtol=2e-5, atol=1e-3)


def test_zq_variants_with_cell_ordering(eng):
    """The optional internal cell ordering (CountTiles sort_rows) is invisible at the API: the ZI / sparse
    loop nests on tiles packed with it give the oracle's sums (general D_hat gathered as the side matrix)."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(31)
    n, m, K = 530, 300, 9
    X = (_rand_counts(rng, n, m, 0.25) * (rng.random((n, 1)) < 0.7)).astype(np.float32)    # 30 % empty cells
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    D = rng.random((n, m)).astype(np.float32)
    ps = rng.random((m, K)); St = (ps > 0.4).astype(np.float32); Sh = ps.astype(np.float32)
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    ct = eng.CountTiles.from_dense(c(X), 'cuda', side=c(D), sort_rows=True)
    assert ct.row_perm is not None
    ws = eng.ZWorkspace(ct, K)
    o = [torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')]
    eng.zq(ws, o[0], o[1], o[2], c(lu), c(lv), S_tilde=c(St), S_hat=c(Sh), w_nz=ct.side_nz)
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    co.zq_sparse_zigap(r[0], r[1], r[2], lu, lv, St, Sh, D, X)
    for got, ref in zip(o, r):
        assert err_colrel(got.cpu().numpy(), ref) < RTOL


@pytest.mark.parametrize('K', [7, 64, 100])
def test_zq_sparse_dead_genes(eng, K):
    """Genes whose mask S_tilde is entirely off (no active factor, sparse_gap.py:113): the reference gets
    den == 0 -> 1 and exactly zero contributions (sparse_gap.py:88-93).  The kernels skip them without the
    slow path -- except where exp(lu + lv) could overflow (inf * 0 = NaN in the reference), which stays
    on the exact path: rows with a huge shift, dead genes with a huge log."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(41 + K)
    n, m = 300, 520
    X = _rand_counts(rng, n, m, 0.3).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    ps = rng.random((m, K)); St = (ps > 0.3).astype(np.float32); Sh = ps.astype(np.float32)
    dead = rng.choice(m, size=60, replace=False)
    St[dead] = 0.0
    lu[5] += 60.0                        # a row outside the shifted form: exact path, also against dead genes
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    ct = eng.CountTiles.from_dense(c(X), 'cuda')
    ws = eng.ZWorkspace(ct, K)
    o = [torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')]
    eng.zq(ws, o[0], o[1], o[2], c(lu), c(lv), S_tilde=c(St), S_hat=c(Sh))
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    co.zq_sparse_gap(r[0], r[1], r[2], lu, lv, St, Sh, X)
    for got, ref in zip(o, r):
        assert np.isfinite(ref).all()
        assert err_colrel(got.cpu().numpy(), ref) < RTOL
    # the dead genes contribute exact (+)zeros; only the shifted row's tiles are flagged
    zj = o[1].cpu().numpy()
    assert not zj[dead].any() and not np.signbit(zj[dead]).any()
    assert 0 < int(ws.tile_flag.sum().item()) <= ct.ncb
    lu2 = lu.copy(); lu2[5] -= 60.0
    eng.zq(ws, o[0], o[1], o[2], c(lu2), c(lv), S_tilde=c(St), S_hat=c(Sh))
    assert int(ws.tile_flag.sum().item()) == 0
    # A dead gene whose exponentials overflow: the reference forms inf * 0 = NaN there (sparse_gap.py:88).
    # It is not skipped: its stored entries take the exact path and give the reference's NaN.  (The
    # reference's dense loop also turns 0 * NaN into NaN for the ZERO counts of such a gene; entries with
    # x == 0 are never visited here -- DESIGN.md section 2.)
    lv3 = lv.copy(); lv3[dead[0], 0] = 95.0
    eng.zq(ws, o[0], o[1], o[2], c(lu2), c(lv3), S_tilde=c(St), S_hat=c(Sh))
    assert int(ws.tile_flag.sum().item()) > 0
    zi = o[0].cpu().numpy()
    hit = X[:, dead[0]] != 0
    assert hit.any() and np.isnan(zi[hit, 0]).all() and np.isfinite(zi[~hit]).all()


def test_zq_gap_every_padding_class(eng):
    """Every kernel configuration the dispatcher can pick (Kp = 16 t or 16 t + 4, lanes per row 4 / 8 / 16,
    column sub-tiles for Kp > 128) and the K values that pad up to it: K = 1 ... 40 and the class
    boundaries up to 256, on a two-tile matrix, against the oracle."""
    rng = np.random.default_rng(99)
    n, m = 290, 270
    X = _rand_counts(rng, n, m, 0.2)
    ks = list(range(1, 41)) + [47, 48, 49, 52, 53, 63, 65, 68, 69, 96, 99, 101, 112, 113, 116, 117, 127, 129, 132,
                               133, 160, 164, 165, 196, 199, 228, 229, 252, 253, 255]
    for K in ks:
        lu = rng.normal(size=(n, K)).astype(np.float32)
        lv = rng.normal(size=(m, K)).astype(np.float32)
        Zi, Zj, rZi, rZj, ws = _run_gap(eng, X, lu, lv)
        assert err_colrel(Zi, rZi) < RTOL, K
        assert err_colrel(Zj, rZj) < RTOL, K


def test_zq_gap_slow_path(eng):
    """Rows / columns outside the range the shifted form covers: exp underflow to denormals,
    den == 0 guard (gap.py:76), huge shifts.  All must land on the reference's float32 answers."""
    rng = np.random.default_rng(5)
    n, m, K = 300, 280, 7
    X = _rand_counts(rng, n, m, 0.3)
    lu = rng.normal(size=(n, K)).astype(np.float32)
    lv = rng.normal(size=(m, K)).astype(np.float32)
    lu[3, :] = -1e15                      # digamma(1e-15): whole row underflows -> den == 0 guard
    lu[7, :] -= 95.0                      # denormal exponentials, den > 0 but tiny
    lu[11, :2] = -1e15                    # partially dead row (fast path: zeros in the factor)
    lu[20, :] += 40.0                     # large shift -> slow path by |mu| test
    lv[5, :] = -1e15
    lv[9, :] -= 60.0
    lv[13, :] += 30.0
    lu[30, :] = np.float32(-80.0); lv[31, :] = np.float32(-30.0)
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, X, lu, lv)
    assert int(ws.tile_flag.sum().item()) > 0
    assert np.isfinite(Zi).all() and np.isfinite(Zj).all()
    assert err_colrel(Zi, rZi) < RTOL
    assert err_colrel(Zj, rZj) < RTOL
    assert not Zi[3].any() and not Zj[5].any()
    _same_zeros(Zi, rZi)
    _same_zeros(Zj, rZj)


def test_zq_gap_slow_path_many_tiles(eng):
    """[r6] From 4096 tiles on oriana_fixup gives a work-group several tiles (nt / 2048, at most 64) and the group walks the
    flagged ones: 65 x 65 tiles with ragged edges, a few hundred rows and genes outside the fast path's range so that more
    than a thousand tiles are flagged -- most groups hold two -- against the oracle's zero-skipping loop nest."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(61)
    n, m, K = 16400, 16500, 5
    X = np.zeros((n, m), np.float32)
    nnz = 600000
    X[rng.integers(0, n, nnz), rng.integers(0, m, nnz)] = rng.integers(1, 40, nnz).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32)
    lv = rng.normal(size=(m, K)).astype(np.float32)
    rows = rng.choice(n, 400, replace=False)
    cols = rng.choice(m, 300, replace=False)
    lu[rows[:200]] -= 95.0                 # denormal exponentials
    lu[rows[200:]] += 40.0                 # large shifts
    lv[cols[:150]] -= 60.0
    lv[cols[150:]] = -1e15                 # den == 0 guard
    ct = eng.CountTiles.from_dense(X, 'cuda')
    assert ct.nrb * ct.ncb >= 4096
    ws = eng.ZWorkspace(ct, K)
    Zi = torch.empty(n, K, dtype=torch.float32, device='cuda')
    Zj = torch.empty(m, K, dtype=torch.float32, device='cuda')
    eng.zq_gap(ws, Zi, Zj, torch.from_numpy(lu).cuda(), torch.from_numpy(lv).cuda())
    torch.cuda.synchronize()
    assert int(ws.tile_flag[:ct.nrb * ct.ncb].sum().item()) > 1000
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    co.zq_gap_nz(rZi, rZj, lu, lv, X)
    Zi, Zj = Zi.cpu().numpy(), Zj.cpu().numpy()
    assert np.isfinite(Zi).all() and np.isfinite(Zj).all()
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL
    _same_zeros(Zi, rZi)
    _same_zeros(Zj, rZj)
    del ct, ws
    torch.cuda.empty_cache()


@pytest.mark.parametrize('K', [20, 50, 100])
def test_den_threshold_follows_the_sums(eng, K):
    """[r4] Factors that have drifted along U c, V / c and specialised (what ZI-pCMF at configs[2] looks like after 25 sweeps):
    every cell's dominant factor sits 30 above the rest, every gene's 25 above the rest, the sums mu_i + mv_j stay around +15.
    Wherever the two dominant factors differ the shifted den' is about 1e-11 -- below the worst-case constant 1e-10, far above
    what the reference's own float32 den needs to be a normal number.  The threshold of the row kernels follows the smallest
    sum of the inputs at hand: no tile goes down the slow path, and the results are the reference's."""
    import os
    if os.environ.get('ORIANA_DEN_THRESHOLD') == 'fixed':
        pytest.skip('the constant threshold is selected')
    rng = np.random.default_rng(K)
    n, m = 300, 280
    X = _rand_counts(rng, n, m, 0.3)
    ku, kv = rng.integers(0, K, size=n), rng.integers(0, K, size=m)
    lu = (5.0 + rng.normal(size=(n, K))).astype(np.float32)
    lv = (-45.0 + rng.normal(size=(m, K))).astype(np.float32)
    lu[np.arange(n), ku] += 30.0
    lv[np.arange(m), kv] += 25.0
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, X, lu, lv)
    thr = float(ws.stats[7])
    lo = float(lu.max(1).min()) + float(lv.max(1).min())
    assert 5.0 < lo < 20.0 and thr == np.float32(1e-25)           # 3e-30 exp(-lo) is below the floor
    assert int(ws.tile_flag.sum().item()) == 0
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL
    # the same factors 60 lower: sums around -45, where the reference's own den runs out of float32 range -- the constant again,
    # and the entries with differing dominant factors take the exact path
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, X, lu - 30.0, lv - 30.0)
    assert float(ws.stats[7]) == np.float32(1e-10)
    assert int(ws.tile_flag.sum().item()) > 0
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL
    # in between the threshold is 3e-30 exp(-sum_lo)
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, X, lu - 25.0, lv - 22.0)
    lo = float((lu - 25.0).max(1).min()) + float((lv - 22.0).max(1).min())
    assert abs(float(ws.stats[7]) / (3e-30 * np.exp(-lo)) - 1.0) < 1e-3 and 1e-25 < float(ws.stats[7]) < 1e-10
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL


def test_zq_gap_rejects_wrong_dtype(eng):
    X = np.ones((4, 4), np.int64)
    ct = eng.CountTiles.from_dense(X, 'cuda')
    ws = eng.ZWorkspace(ct, 2)
    z = torch.zeros(4, 2, device='cuda')
    with pytest.raises(TypeError):
        eng.zq_gap(ws, z, z.clone(), z.double(), z.clone())


def test_stateless_dropin_signature(eng):
    """GaP.compute_Z_q_expectations(Z_i, Z_j, log_U, log_V, X): the reference's own calling
    convention (gap.py:89-94) on device tensors, through oriana_zq_gap_f32."""
    from oriana_amd.models import GaP
    from oracle import cavi_oracle as co
    g = load_golden(golden_files('gap_odd_nmf.npz')[0])
    X = torch.from_numpy(g['X'].astype(np.float32)).cuda()
    lu = torch.from_numpy(g['s0/log_U_hat']).cuda(); lv = torch.from_numpy(g['s0/log_V_hat']).cuda()
    Zi = torch.full(lu.shape, 7.0, device='cuda'); Zj = torch.full(lv.shape, 7.0, device='cuda')   # callee zero-fills
    assert GaP.compute_Z_q_expectations(Zi, Zj, lu, lv, X) is None
    assert err_colrel(Zi.cpu().numpy(), g['kernel/Zi']) < RTOL
    assert err_colrel(Zj.cpu().numpy(), g['kernel/Zj']) < RTOL
    with pytest.raises(TypeError):
        GaP.compute_Z_q_expectations(Zi, Zj, lu.double(), lv, X)
    # all-zero and empty inputs
    Z0 = torch.zeros_like(X)
    GaP.compute_Z_q_expectations(Zi, Zj, lu, lv, Z0)
    assert not Zi.any() and not Zj.any()


def _variant_inputs(g):
    s0 = {k[3:]: v for k, v in g.items() if k.startswith('s0/')}
    name = str(g['meta/name'])
    out = dict(lu=s0['log_U_hat'], lv=s0['log_V_hat'], D=None, St=None, Sh=None)
    if 'p_d' in s0:
        out['D'] = s0['p_d'].astype(np.float32)
    if 'p_s' in s0:
        out['St'] = (s0['p_s'] > float(g['meta/tau'])).astype(np.float32)
        out['Sh'] = s0['p_s'].astype(np.float32)
    return name, out


@pytest.mark.parametrize('path', golden_files('zigap_*.npz') + golden_files('sparse*.npz'), ids=os.path.basename)
def test_zq_variants_golden(eng, path):
    """ZIGaP / SparseGaP / SparseZIGaP loop nests (zigap.py:79-95, sparse_gap.py:81-97,
    sparse_zigap.py:100-116) against the reference's own outputs on the post-init state."""
    g = load_golden(path)
    name, a = _variant_inputs(g)
    X = g['X']; n, m = X.shape; K = a['lu'].shape[1]
    if a['D'] is not None:
        assert (a['D'][X != 0] == 1.0).all()          # what the model path relies on (zigap.py:135)
    ct = eng.CountTiles.from_dense(X, 'cuda')
    ws = eng.ZWorkspace(ct, K)
    dev = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v)).cuda()
    Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda'); Zl = torch.empty(m, K, device='cuda')
    dq = dev(a['D'][:, :K]) if name == 'ZIGaP' else None      # zigap.py:94 reads D_hat[i, k]
    eng.zq(ws, Zi, Zj, Zl, dev(a['lu']), dev(a['lv']), S_tilde=dev(a['St']), S_hat=dev(a['Sh']), dq=dq)
    torch.cuda.synchronize()
    assert err_colrel(Zi.cpu().numpy(), g['kernel/Zi']) < RTOL
    assert err_colrel(Zj.cpu().numpy(), g['kernel/Zj']) < RTOL
    assert err_colrel(Zl.cpu().numpy(), g['kernel/Zlog']) < RTOL
    assert np.isfinite(Zl.cpu().numpy()).all()


def test_zq_zigap_without_quirk(eng):
    """reference_quirks=False: the evident D_hat[i, j] in zigap.py:94 (oracle quirk=False)."""
    from oracle import cavi_oracle as co
    g = load_golden(golden_files('zigap_odd_rand.npz')[0])
    name, a = _variant_inputs(g)
    X = g['X']; n, m = X.shape; K = a['lu'].shape[1]
    ct = eng.CountTiles.from_dense(X, 'cuda')
    ws = eng.ZWorkspace(ct, K)
    Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda'); Zl = torch.empty(m, K, device='cuda')
    eng.zq(ws, Zi, Zj, Zl, torch.from_numpy(a['lu']).cuda(), torch.from_numpy(a['lv']).cuda())
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    co.zq_zigap(r[0], r[1], r[2], a['lu'], a['lv'], a['D'], np.ascontiguousarray(X.astype(np.float32)), quirk=False)
    for got, ref in zip((Zi, Zj, Zl), r):
        assert err_colrel(got.cpu().numpy(), ref) < RTOL


@pytest.mark.parametrize('K', [7, 100, 200])
def test_zq_sparse_random(eng, K):
    """Sparse loop nest on random inputs with inactive factors (S_tilde zeros), several tiles."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(K)
    n, m = 300, 520
    X = _rand_counts(rng, n, m, 0.1)
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    ps = rng.random((m, K))
    St = (ps > 0.3).astype(np.float32); Sh = ps.astype(np.float32)
    St[5] = 0                                   # a gene with every factor switched off: den == 0 guard
    ct = eng.CountTiles.from_dense(X, 'cuda')
    ws = eng.ZWorkspace(ct, K)
    Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda'); Zl = torch.empty(m, K, device='cuda')
    c = lambda v: torch.from_numpy(v).cuda()
    eng.zq(ws, Zi, Zj, Zl, c(lu), c(lv), S_tilde=c(St), S_hat=c(Sh))
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    co.zq_sparse_gap(r[0], r[1], r[2], lu, lv, St, Sh, np.ascontiguousarray(X.astype(np.float32)))
    for got, ref in zip((Zi, Zj, Zl), r):
        assert err_colrel(got.cpu().numpy(), ref) < RTOL
    assert not Zj.cpu().numpy()[5].any()


@pytest.mark.parametrize('K,weighted', [(3, False), (20, True), (50, False), (64, True), (64, False), (70, False)])
def test_sparse_row_phase_fused_matches_split(eng, K, weighted, monkeypatch):
    """The sparse loop nest with the S_hat-weighted row sums folded into the row pass and the two per-gene sums taken in
    one column pass (two factor images in LDS, Kp <= 64) against the four-kernel form (row pass + row product over the
    stored s, two column passes), and both against the oracle; with slow-path entries and a dead gene; K = 70 does not
    fit two images and takes the four-kernel form in both runs."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(K + 17)
    n, m = 700, 530
    X = _rand_counts(rng, n, m, 0.12)
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    lu[3] -= 80.0                               # a cell whose entries take the exact slow path
    ps = rng.random((m, K))
    St = (ps > 0.3).astype(np.float32); Sh = ps.astype(np.float32)
    St[7] = 0                                   # a gene with every factor switched off
    ct = eng.CountTiles.from_dense(X, 'cuda')
    c = lambda v: torch.from_numpy(v).cuda()
    D = rng.random((n, m)).astype(np.float32) if weighted else None
    w_nz = None
    if weighted:
        ct = eng.CountTiles.from_dense(X, 'cuda', side=c(D))
        w_nz = ct.side_nz
    out = {}
    for mode in (True, False):
        monkeypatch.setattr(eng, '_FUSE_SPARSE_ROWS', mode)
        monkeypatch.setattr(eng, '_FUSE_SPARSE_COLS', mode)     # (same for the two per-gene sums: one column pass / two)
        ws = eng.ZWorkspace(ct, K)
        Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda'); Zl = torch.empty(m, K, device='cuda')
        eng.zq(ws, Zi, Zj, Zl, c(lu), c(lv), S_tilde=c(St), S_hat=c(Sh), w_nz=w_nz)
        out[mode] = [t.cpu().numpy() for t in (Zi, Zj, Zl)]
        assert (ws.s_rs is None) == (mode and K <= 64)      # the fused form never needs the row-side copy of s
    for a, b in zip(out[True], out[False]):
        assert err_colrel(a, b) < 2e-6
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    Xf = np.ascontiguousarray(X.astype(np.float32))
    if weighted:
        co.zq_sparse_zigap(r[0], r[1], r[2], lu, lv, St, Sh, D, Xf)
    else:
        co.zq_sparse_gap(r[0], r[1], r[2], lu, lv, St, Sh, Xf)
    for got, ref in zip(out[True], r):
        assert err_colrel(got, ref) < RTOL


@pytest.mark.parametrize('name', ['ZIGaP', 'SparseGaP', 'SparseZIGaP'])
def test_variant_dropins_general_D(eng, name):
    """Model.compute_Z_q_expectations(...) with the reference's argument order on dense device
    tensors, with a GENERAL D_hat (not 1 at the non-zeros): exercises the weighted kernels."""
    import oriana_amd.models as M
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(11)
    n, m, K = 300, 280, 20
    X = _rand_counts(rng, n, m, 0.2).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    lu[3] = -1e15; lv[7] -= 70.0                       # slow-path entries too
    D = rng.random((n, m)).astype(np.float32)
    ps = rng.random((m, K)); St = (ps > 0.4).astype(np.float32); Sh = ps.astype(np.float32)
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    o = [torch.full((n, K), 3.0, device='cuda'), torch.full((m, K), 3.0, device='cuda'), torch.full((m, K), 3.0, device='cuda')]
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    if name == 'ZIGaP':
        assert M.ZIGaP.compute_Z_q_expectations(o[0], o[1], o[2], c(lu), c(lv), c(D), c(X)) is None
        co.zq_zigap(r[0], r[1], r[2], lu, lv, D, X, quirk=True)
    elif name == 'SparseGaP':
        M.SparseGaP.compute_Z_q_expectations(o[0], o[1], o[2], c(lu), c(lv), c(St), c(Sh), c(X))
        co.zq_sparse_gap(r[0], r[1], r[2], lu, lv, St, Sh, X)
    else:
        M.SparseZIGaP.compute_Z_q_expectations(o[0], o[1], o[2], c(lu), c(lv), c(St), c(Sh), c(D), c(X))
        co.zq_sparse_zigap(r[0], r[1], r[2], lu, lv, St, Sh, D, X)
    for got, ref in zip(o, r):
        assert err_colrel(got.cpu().numpy(), ref) < RTOL


def test_zigap_dropin_corrected_index_and_edges(eng):
    """oriana_zq_zigap_f32 with reference_quirks = 0 (per-gene sums weighted by D_hat[i, j]) against
    the oracle's corrected form; all-zero X and empty shapes return zero-filled outputs."""
    import oriana_amd.models as M
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(5)
    n, m, K = 270, 131, 7
    X = _rand_counts(rng, n, m, 0.3).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    D = rng.random((n, m)).astype(np.float32)
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    o = [torch.full((n, K), 3.0, device='cuda'), torch.full((m, K), 3.0, device='cuda'), torch.full((m, K), 3.0, device='cuda')]
    r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
    M.ZIGaP.compute_Z_q_expectations(o[0], o[1], o[2], c(lu), c(lv), c(D), c(X), reference_quirks=False)
    co.zq_zigap(r[0], r[1], r[2], lu, lv, D, X, quirk=False)
    for got, ref in zip(o, r):
        assert err_colrel(got.cpu().numpy(), ref) < RTOL
    # all-zero counts: outputs are zero-filled by the callee
    M.ZIGaP.compute_Z_q_expectations(o[0], o[1], o[2], c(lu), c(lv), c(D), c(np.zeros_like(X)))
    assert all(not t.cpu().numpy().any() for t in o)
    # the quirk needs K <= m (zigap.py:94 would raise IndexError)
    with pytest.raises(Exception):
        M.ZIGaP.compute_Z_q_expectations(torch.zeros(4, 9, device='cuda'), torch.zeros(3, 9, device='cuda'),
                                         torch.zeros(3, 9, device='cuda'), torch.zeros(4, 9, device='cuda'),
                                         torch.zeros(3, 9, device='cuda'), torch.ones(4, 3, device='cuda'),
                                         torch.ones(4, 3, device='cuda'))


def test_pack_from_scipy_matches_dense(eng):
    """CountTiles.from_scipy (row chunks of a CSR matrix expanded on the device) builds the same
    tiles as from_dense; a model built from a sparse CountMatrix steps like the dense one."""
    import scipy.sparse as sp
    import oriana_amd.models as M
    from oriana_amd.singlecell import CountMatrix
    rng = np.random.default_rng(23)
    n, m, K = 700, 300, 6
    X = _rand_counts(rng, n, m, 0.15).astype(np.float32)
    X[300:560] = 0                                            # an empty row block
    A = sp.csr_matrix(X)
    ct_d = eng.CountTiles.from_dense(X, 'cuda')
    ct_s = eng.CountTiles.from_scipy(A, 'cuda', chunk_rows=256)
    assert ct_s.nnz == ct_d.nnz == A.nnz
    assert np.array_equal(ct_s.to_dense(), X) and np.array_equal(ct_d.to_dense(), X)
    a1 = rng.gamma(1.0, 1.0, size=(n, K)); b1 = rng.gamma(1.0, 1.0, size=(m, K))
    g_d = M.GaP(X, k=K, init=(a1, b1)); g_s = M.GaP(CountMatrix(A), k=K, init=(a1, b1))
    g_d.fit(2); g_s.fit(2)
    for k, v in g_d.state().items():
        assert err_colrel(g_s.state()[k], v) < 1e-6, k
    assert sp.issparse(CountMatrix(A).as_sparse_matrix('csr'))


# ---- dense f64 products of the ZI models on the matrix cores (csrc/dense_mfma.hip) ------------------

@pytest.mark.gpu
@pytest.mark.parametrize('n,m,K', [(1, 1, 1), (37, 53, 3), (300, 517, 20), (1000, 260, 50), (513, 1024, 100),
                                   (260, 300, 130), (70, 90, 256), (5000, 3001, 50), (300, 260, 40), (260, 300, 70),
                                   (300, 300, 90), (270, 310, 120)])
@pytest.mark.parametrize('trans', [0, 1])
def test_dense_times_factor(n, m, K, trans):
    import torch
    from oriana_amd._lib import call, ptr, stream_ptr
    g = torch.Generator(device='cpu').manual_seed(n * 7 + m * 3 + K + trans)
    D = torch.rand(n, m, generator=g, dtype=torch.float32)
    D[D < 0.3] = 0.0
    W = torch.rand(n if trans else m, K, generator=g, dtype=torch.float64) * 3.0
    Dd, Wd = D.cuda(), W.cuda()
    out = torch.zeros(m if trans else n, K, dtype=torch.float64, device='cuda')
    call('oriana_dense_times_factor', ptr(out), ptr(Dd), ptr(Wd), n, m, K, trans, stream_ptr())
    torch.cuda.synchronize()
    ref = (D.double().t() @ W) if trans else (D.double() @ W)
    # f64 products, summation order differs from BLAS: 1e-12 relative
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-12, atol=1e-12)
    # accumulating call: out += product
    call('oriana_dense_times_factor', ptr(out), ptr(Dd), ptr(Wd), n, m, K, trans, stream_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), 2.0 * ref.numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('n,m,K', [(1, 1, 1), (37, 53, 3), (300, 517, 20), (1000, 260, 50), (513, 1024, 100),
                                   (70, 90, 256), (2000, 1500, 0)])
def test_dropout_update_fused_matches_unfused(n, m, K):
    """The fused MFMA kernel against the two-step path (f64 GEMM + oriana_dropout_update)."""
    import torch
    from oriana_amd._lib import call, ptr, stream_ptr
    g = torch.Generator(device='cpu').manual_seed(n + 3 * m + 11 * K)
    U = (torch.rand(n, K, generator=g, dtype=torch.float64) * 2.0).cuda()
    V = (torch.rand(m, K, generator=g, dtype=torch.float64) * 2.0).cuda()
    pi = torch.rand(m, generator=g, dtype=torch.float64)
    if m > 4:
        pi[1] = 0.0
        pi[3] = 1.0
    pi = pi.cuda()
    X = (torch.rand(n, m, generator=g) < 0.2).float().cuda()
    mask = torch.zeros(((n + 31) // 32) * m, dtype=torch.int32, device='cuda')
    call('oriana_nzmask_f32', ptr(mask), ptr(X), n, m, stream_ptr())
    p1 = torch.empty(n, m, dtype=torch.float64, device='cuda')
    D1 = torch.empty(n, m, dtype=torch.float32, device='cuda')
    cs1 = torch.zeros(m, dtype=torch.float64, device='cuda')
    call('oriana_dropout_update_fused', ptr(p1), ptr(D1), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(cs1), n, m, K,
         stream_ptr())
    lam = U @ V.t() if K else torch.zeros(n, m, dtype=torch.float64, device='cuda')
    p2 = torch.empty_like(p1)
    D2 = torch.empty_like(D1)
    cs2 = torch.zeros_like(cs1)
    call('oriana_dropout_update', ptr(p2), ptr(D2), ptr(lam.contiguous()), ptr(pi), ptr(mask), ptr(cs2), n, m,
         stream_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(p1.cpu().numpy(), p2.cpu().numpy(), rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(D1.cpu().numpy(), D2.cpu().numpy(), rtol=2e-7)
    np.testing.assert_allclose(cs1.cpu().numpy(), cs2.cpu().numpy(), rtol=1e-11)
    # the overrides are exact
    p1h, Xh = p1.cpu().numpy(), X.cpu().numpy()
    assert np.all(p1h[Xh != 0] == 1.0 - 1e-10)
    if m > 4:
        assert np.all(p1h[:, 1][Xh[:, 1] == 0] == 1e-10)
        assert np.all(p1h[:, 3] == 1.0 - 1e-10)


# ---- the dense work of a ZI sweep on the float32 matrix cores (csrc/dense_f32.hip) -------------------------------

F32_SHAPES = [(1, 1, 1), (37, 53, 3), (300, 517, 20), (1000, 260, 50), (513, 1024, 100), (129, 2050, 33),
              (2000, 70, 64), (260, 300, 97), (70, 90, 128), (5000, 3001, 50), (4100, 600, 50),
              # 32 < K <= 100 with a gene count that is a multiple of 4: csrc/dense_zi.hip (every (KC, TAIL) pair, partial
              # cell and gene tiles, fewer cells than a tile, several gene ranges / cell ranges per work-group column)
              (33, 36, 65), (3000, 2080, 84), (700, 5000, 96), (257, 132, 100), (9000, 420, 80), (31, 4, 100),
              (513, 1022, 100), (300, 520, 40), (2000, 132, 48), (77, 64, 33), (640, 96, 52), (700, 520, 68),
              # Kp = 64 (round 6: the D update on k_zi_row<4, 0>, D^T U on k_zi_col<4, 0>), K = 49 .. 52 with the tail in the last tile
              (2100, 520, 64), (300, 132, 57), (5000, 2080, 60), (1030, 1028, 49), (290, 260, 51)]


@pytest.mark.gpu
@pytest.mark.parametrize('n,m,K', F32_SHAPES)
@pytest.mark.parametrize('arithmetic', [0, 1], ids=['f32', 'bf16x3'])
def test_dense_t_times_factor_f32(n, m, K, arithmetic):
    """D_hat^T U_hat (zigap.py:124) with float32 products (the float32 matrix instruction, or three-way bf16 splits on the
    bf16 matrix cores for K <= 100) and sums that end in float64, against the float64 product."""
    import torch
    from oriana_amd import _lib
    from oriana_amd._lib import call, ptr, stream_ptr
    g = torch.Generator(device='cpu').manual_seed(n * 7 + m * 3 + K)
    D = torch.rand(n, m, generator=g, dtype=torch.float32)
    D[D < 0.3] = 0.0
    D[torch.rand(n, m, generator=g) < 0.05] *= 1e-6          # some tiny dropout probabilities
    W = torch.rand(n, K, generator=g, dtype=torch.float64) * 3.0
    Dd, Wd = D.cuda(), W.cuda()
    out = torch.zeros(m, K, dtype=torch.float64, device='cuda')
    scr = torch.zeros(int(_lib.load().oriana_dense_t_scratch_floats(n, K)), dtype=torch.float32, device='cuda')
    call('oriana_dense_t_times_factor_f32', ptr(out), ptr(Dd), ptr(Wd), ptr(scr), arithmetic, n, m, K, stream_ptr())
    torch.cuda.synchronize()
    ref = (D.double().t() @ W).numpy()
    # positive terms: relative error of the sums.  256-term float32 chains (3e-7 rms each) averaged over the chunks
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-30)
    if n >= 1000:
        rel = (out.cpu().numpy() - ref) / np.maximum(ref, 1e-300)
        assert np.sqrt(np.mean(rel ** 2)) < 2.5e-7
        assert abs(np.mean(rel)) < 1e-7                        # no drift: the matrix-core sums are cut every 256 terms
    call('oriana_dense_t_times_factor_f32', ptr(out), ptr(Dd), ptr(Wd), ptr(scr), arithmetic, n, m, K, stream_ptr())   # accumulates
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), 2.0 * ref, rtol=1e-6, atol=1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize('n,m,K', F32_SHAPES)
@pytest.mark.parametrize('with_next', [True, False])
@pytest.mark.parametrize('arithmetic', [0, 1], ids=['f32', 'bf16x3'])
def test_dropout_sweep_fused_matches_float64(n, m, K, with_next, arithmetic):
    """oriana_dropout_sweep_fused against the float64 kernels: D_hat, the column sums of p_d, and D_hat V_next, with
    the float32 matrix instruction and with three-way bf16 splits on the bf16 matrix cores (K <= 100)."""
    import torch
    from oriana_amd import _lib
    from oriana_amd._lib import call, ptr, stream_ptr
    g = torch.Generator(device='cpu').manual_seed(n + 3 * m + 11 * K)
    U = (torch.rand(n, K, generator=g, dtype=torch.float64) * 2.0).cuda()
    V = (torch.rand(m, K, generator=g, dtype=torch.float64) * 2.0).cuda()
    Vn = (torch.rand(m, K, generator=g, dtype=torch.float64) * 3.0).cuda()
    pi = torch.rand(m, generator=g, dtype=torch.float64)
    if m > 4:
        pi[1] = 0.0
        pi[3] = 1.0
    pi = pi.cuda()
    X = (torch.rand(n, m, generator=g) < 0.2).float().cuda()
    mask = torch.zeros(((n + 31) // 32) * m, dtype=torch.int32, device='cuda')
    call('oriana_nzmask_f32', ptr(mask), ptr(X), n, m, stream_ptr())
    D1 = torch.full((n, m), -7.0, dtype=torch.float32, device='cuda')
    cs1 = torch.zeros(m, dtype=torch.float64, device='cuda')
    DV = torch.zeros(n, K, dtype=torch.float64, device='cuda')
    lgs = torch.zeros(int(_lib.load().oriana_dropout_sweep_scratch_floats(m, K)), dtype=torch.float32, device='cuda')
    dropout_sweep(D1, U, V, pi, mask, cs1, Vn if with_next else None, DV if with_next else None, lgs, arithmetic, n, m, K)
    D2 = torch.empty_like(D1)
    cs2 = torch.zeros_like(cs1)
    call('oriana_dropout_update_fused', None, ptr(D2), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(cs2), n, m, K, stream_ptr())
    torch.cuda.synchronize()
    d1, d2 = D1.cpu().numpy(), D2.cpu().numpy()
    # Lambda carries K float32 roundings (<= 1e-7 K^0.5 |Lambda| absolute in the exponent), the sigmoid two more ulp
    lam_max = float((U @ V.t()).max())
    # (the six-product bf16 form of a float32 product drops terms below 2^-24 of it: a slightly wider bound there)
    # (float32 instruction: measured 4.02e-7 |Lambda| at K = 60, |Lambda| = 102 -- 0.5 ulp x K^0.5 would be 4.6e-7)
    assert np.max(np.abs(d1 - d2) / np.maximum(d2, 1e-30) * (d2 > 1e-30)) < (6e-7 if arithmetic else 5e-7) * max(1.0, lam_max)
    np.testing.assert_allclose(d1, d2, rtol=0, atol=3e-7)
    Xh = X.cpu().numpy()
    assert np.all(d1[Xh != 0] == 1.0)                                     # the overrides are exact
    if m > 4:
        assert np.all(d1[:, 1][Xh[:, 1] == 0] == np.float32(1e-10))
        assert np.all(d1[:, 3] == 1.0)
    np.testing.assert_allclose(cs1.cpu().numpy(), cs2.cpu().numpy(), rtol=3e-7)
    if with_next:
        ref = (D1.double() @ Vn).cpu().numpy()                            # the product of the D_hat it stored
        np.testing.assert_allclose(DV.cpu().numpy(), ref, rtol=1e-6, atol=1e-30)
        if m >= 1000:
            rel = (DV.cpu().numpy() - ref) / np.maximum(ref, 1e-300)
            assert np.sqrt(np.mean(rel ** 2)) < 2.5e-7
            assert abs(np.mean(rel)) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize('arithmetic', [0, 1], ids=['f32', 'bf16x3'])
def test_dense_sweep_entries_wide_dynamic_range(arithmetic):
    """Operands spread over 15 decades (Gamma means after many sweeps: a few dominant factors, the rest at the
    1e-15 clamp): the three-way bf16 split keeps the float32 exponent range, so both arithmetics stay at float32
    accuracy on every sum."""
    import torch
    from oriana_amd import _lib
    from oriana_amd._lib import call, ptr, stream_ptr
    n, m, K = 1500, 1100, 50
    g = torch.Generator(device='cpu').manual_seed(5)
    def logu(shape, lo, hi):
        return torch.exp(torch.rand(shape, generator=g, dtype=torch.float64) * (np.log(hi) - np.log(lo)) + np.log(lo))
    U = logu((n, K), 1e-15, 3.0).cuda(); V = logu((m, K), 1e-15, 3.0).cuda(); Vn = logu((m, K), 1e-12, 1e3).cuda()
    pi = torch.rand(m, generator=g, dtype=torch.float64).cuda()
    D1 = torch.empty(n, m, dtype=torch.float32, device='cuda'); D2 = torch.empty_like(D1)
    cs = torch.zeros(m, dtype=torch.float64, device='cuda')
    DV = torch.zeros(n, K, dtype=torch.float64, device='cuda')
    scr = torch.zeros(int(_lib.load().oriana_dropout_sweep_scratch_floats(m, K)), dtype=torch.float32, device='cuda')
    # (a mask of zeros: without the per-lane flags the K = 33 .. 100 kernel of csrc/dense_zi.hip would not run)
    zmask = torch.zeros(((n + 31) // 32) * m, dtype=torch.int32, device='cuda')
    dropout_sweep(D1, U, V, pi, zmask, cs, Vn, DV, scr, arithmetic, n, m, K)
    call('oriana_dropout_update_fused', None, ptr(D2), ptr(U), ptr(V), ptr(pi), None, None, n, m, K, stream_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(D1.cpu().numpy(), D2.cpu().numpy(), rtol=2e-5, atol=3e-7)
    ref = (D1.double() @ Vn).cpu().numpy()
    np.testing.assert_allclose(DV.cpu().numpy(), ref, rtol=1e-6)
    W = logu((n, K), 1e-15, 1e3).cuda()
    out = torch.zeros(m, K, dtype=torch.float64, device='cuda')
    s2 = torch.zeros(int(_lib.load().oriana_dense_t_scratch_floats(n, K)), dtype=torch.float32, device='cuda')
    call('oriana_dense_t_times_factor_f32', ptr(out), ptr(D1), ptr(W), ptr(s2), arithmetic, n, m, K, stream_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), (D1.double().t() @ W).cpu().numpy(), rtol=1e-6)


@pytest.mark.gpu
def test_f32_dense_entries_reject_large_K():
    import torch
    from oriana_amd import _lib
    from oriana_amd._lib import ptr
    t = torch.zeros(4, device='cuda', dtype=torch.float64)
    f = torch.zeros(4, device='cuda', dtype=torch.float32)
    L = _lib.load()
    assert L.oriana_dense_t_times_factor_f32(ptr(t), ptr(f), ptr(t), None, 0, 1, 1, 129, None) == -2
    assert L.oriana_dense_t_times_factor_f32(ptr(t), ptr(f), ptr(t), None, 1, 1, 1, 3, None) == -1      # bf16x3 needs its scratch
    assert L.oriana_dropout_sweep_fused(ptr(f), ptr(t), ptr(t), ptr(t), None, None, None, None, ptr(f), 0, 1, 1, 129, None) == -2
    assert L.oriana_dropout_sweep_fused(ptr(f), ptr(t), ptr(t), ptr(t), None, None, None, None, ptr(f), 7, 1, 1, 1, None) == -1


@pytest.mark.parametrize('K,m', [(100, 700), (96, 300), (20, 530), (64, 513), (200, 300)])
def test_deterministic_column_pass(eng, K, m):
    """Debug mode of SURVEY.md section 5 (engine.set_deterministic): the per-gene sums are combined in a fixed order
    (per work item slabs + ordered reduction) instead of float atomics: two runs are bit-identical, and the default
    (atomic) path agrees with it to the order of float32 additions.  K = 96 / 100 take the two-tile kernel (odd and
    even numbers of column tiles), the others the generic one (G = 4, G = 8 with two work-groups per item)."""
    rng = np.random.default_rng(K + m)
    n = 1500
    dens = rng.beta(1.0, 3.0, size=m)
    X = (rng.poisson(3.0, size=(n, m)) + 1) * (rng.random((n, m)) < dens)
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    ct = eng.CountTiles.from_dense(X.astype(np.float32), 'cuda')
    ct._col_work[int(eng._lib.load().oriana_col_block_tiles(K))] = ct._build_col_work(
        target_items=24, width=int(eng._lib.load().oriana_col_block_tiles(K)))      # several items per column block
    ws = eng.ZWorkspace(ct, K)
    tlu, tlv = torch.from_numpy(lu).cuda(), torch.from_numpy(lv).cuda()
    outs = []
    try:
        for det in (True, True, False):
            eng.set_deterministic(det)
            Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda')
            eng.zq_gap(ws, Zi, Zj, tlu, tlv)
            outs.append((Zi.cpu().numpy(), Zj.cpu().numpy()))
    finally:
        eng.set_deterministic(False)
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][0], outs[1][0])
    assert err_colrel(outs[2][1], outs[0][1]) < 1e-6
    from oracle import cavi_oracle as co
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    co.zq_gap(rZi, rZj, lu, lv, np.ascontiguousarray(X.astype(np.float32)))
    assert err_colrel(outs[0][1], rZj) < RTOL and err_colrel(outs[0][0], rZi) < RTOL


@pytest.mark.parametrize('K', [85, 90, 96, 97, 100])
@pytest.mark.parametrize('m', [250, 700])
def test_k100_kernels_shapes(eng, K, m):
    """The K = 85..100 kernels (two lanes per row, duplicated chunk groups, two column tiles per image): Kp = 96
    and 100, one and three column tiles (a column-tile pair with a missing second tile), ragged row blocks."""
    rng = np.random.default_rng(K * 7 + m)
    n = 777
    X = (rng.poisson(2.0, size=(n, m)) + 1) * (rng.random((n, m)) < rng.beta(1.0, 2.0, size=m))
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, X.astype(np.int64), lu, lv)
    assert eng._lib.load().oriana_col_block_tiles(K) == 2
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL
    np.testing.assert_allclose(Zi.sum(1), X.sum(1), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(Zj.sum(1), X.sum(0), rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize('shift_u,shift_v,fast', [(35.0, -30.0, True), (-40.0, 38.0, True), (0.0, 0.0, True),
                                                  (60.0, 30.0, False), (-70.0, -30.0, False)])
def test_scale_drift_keeps_the_fast_path(eng, shift_u, shift_v, fast):
    """CAVI drifts along the scale indeterminacy (U c, V / c): ZI-pCMF at BASELINE configs[2] reaches row maxima of
    E[log U] of +45 and of E[log V] of -27 within 25 sweeps.  Only the SUMS enter the loop nest (gap.py:74), so such
    a state must stay on the shifted form (no flagged tile) and agree with the oracle; sums that really leave the
    float32 range (exp overflow / total underflow in the reference) still go through the exact path and agree too."""
    rng = np.random.default_rng(11)
    n, m, K = 400, 300, 100
    X = (rng.poisson(3.0, size=(n, m)) + 1) * (rng.random((n, m)) < 0.2)
    lu = (rng.normal(size=(n, K)) * 1.5 + shift_u).astype(np.float32)
    lv = (rng.normal(size=(m, K)) * 1.5 + shift_v).astype(np.float32)
    with np.errstate(all='ignore'):
        Zi, Zj, rZi, rZj, ws = _run_gap(eng, X.astype(np.int64), lu, lv)
    nflag = int(ws.tile_flag.sum().item())
    assert (nflag == 0) == fast, nflag
    fin = np.isfinite(rZi)
    assert np.array_equal(np.isfinite(Zi), fin) and np.array_equal(np.isfinite(Zj), np.isfinite(rZj))
    if fin.all():
        # sums near -100: every exp(lu + lv) of the reference is a float32 DENORMAL (a few significant bits, and
        # glibc / the device round them differently): both sides reproduce that arithmetic, to ~1e-3
        tol = RTOL if fast else 2e-3
        assert err_colrel(Zi, rZi) < tol and err_colrel(Zj, rZj) < tol


@pytest.mark.gpu
def test_dense_zi_kernels_decline_unaligned_buffers():
    """csrc/dense_zi.hip moves 16-byte pieces: a D_hat that starts 4 bytes off a 16-byte boundary takes the float32
    instruction's kernels instead (same results to float32 accuracy), through both entries."""
    import torch
    from oriana_amd import _lib
    from oriana_amd._lib import call, ptr, stream_ptr
    n, m, K = 300, 260, 100
    g = torch.Generator(device='cpu').manual_seed(3)
    U = (torch.rand(n, K, generator=g, dtype=torch.float64) * 0.3).cuda()
    V = (torch.rand(m, K, generator=g, dtype=torch.float64) * 0.3).cuda()
    pi = torch.rand(m, generator=g, dtype=torch.float64).cuda()
    X = (torch.rand(n, m, generator=g) < 0.2).float().cuda()
    mask = torch.zeros(((n + 31) // 32) * m, dtype=torch.int32, device='cuda')
    call('oriana_nzmask_f32', ptr(mask), ptr(X), n, m, stream_ptr())
    lgs = torch.zeros(int(_lib.load().oriana_dropout_sweep_scratch_floats(m, K)), dtype=torch.float32, device='cuda')
    dts = torch.zeros(int(_lib.load().oriana_dense_t_scratch_floats(n, K)), dtype=torch.float32, device='cuda')
    outs = []
    for shift in (0, 1):
        base = torch.zeros(n * m + 4, dtype=torch.float32, device='cuda')
        D = base[shift:shift + n * m].view(n, m)
        assert (D.data_ptr() % 16 == 0) == (shift == 0)
        cs = torch.zeros(m, dtype=torch.float64, device='cuda')
        DV = torch.zeros(n, K, dtype=torch.float64, device='cuda')
        dropout_sweep(D, U, V, pi, mask, cs, V, DV, lgs, 1, n, m, K)
        DtU = torch.zeros(m, K, dtype=torch.float64, device='cuda')
        call('oriana_dense_t_times_factor_f32', ptr(DtU), ptr(D), ptr(U), ptr(dts), 1, n, m, K, stream_ptr())
        torch.cuda.synchronize()
        outs.append((D.clone(), cs, DV, DtU))
    for a, b in zip(outs[0], outs[1]):
        assert torch.allclose(a.double(), b.double(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('K', [33, 36, 40, 48, 50, 52, 57, 64])
@pytest.mark.parametrize('m', [250, 700])
def test_k64_kernels_shapes(eng, K, m):
    """The 33 <= Kp <= 64 kernels (two lanes per row / column, image rows zero-padded to 64 floats, two column tiles per
    image): every padded width (36, 48, 52, 64), one and three column tiles (a pair with a missing second tile), ragged
    row blocks -- the pCMF nest, then the sparse nest through the fused row pass (second image) and the dual column
    pass, with a dead gene and per-entry weights."""
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(K * 11 + m)
    n = 777
    X = ((rng.poisson(2.0, size=(n, m)) + 1) * (rng.random((n, m)) < rng.beta(1.0, 2.0, size=m))).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32); lv = rng.normal(size=(m, K)).astype(np.float32)
    Zi, Zj, rZi, rZj, ws = _run_gap(eng, X.astype(np.int64), lu, lv)
    assert err_colrel(Zi, rZi) < RTOL and err_colrel(Zj, rZj) < RTOL
    np.testing.assert_allclose(Zi.sum(1), X.sum(1), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(Zj.sum(1), X.sum(0), rtol=2e-5, atol=1e-3)
    # sparse nest (sparse_gap.py:81-97 / sparse_zigap.py:100-116 with general D_hat)
    ps = rng.random((m, K))
    St = (ps > 0.3).astype(np.float32); Sh = ps.astype(np.float32)
    St[7] = 0
    lu2 = lu.copy(); lu2[3] -= 80.0                    # a cell on the exact slow path
    c = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    for weighted in (False, True):
        D = rng.random((n, m)).astype(np.float32) if weighted else None
        ct = eng.CountTiles.from_dense(c(X), 'cuda', side=(c(D) if weighted else None))
        ws = eng.ZWorkspace(ct, K)
        o = [torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')]
        eng.zq(ws, o[0], o[1], o[2], c(lu2), c(lv), S_tilde=c(St), S_hat=c(Sh), w_nz=(ct.side_nz if weighted else None))
        r = [np.empty((n, K), np.float32), np.empty((m, K), np.float32), np.empty((m, K), np.float32)]
        if weighted:
            co.zq_sparse_zigap(r[0], r[1], r[2], lu2, lv, St, Sh, D, X)
        else:
            co.zq_sparse_gap(r[0], r[1], r[2], lu2, lv, St, Sh, X)
        for got, ref in zip(o, r):
            assert err_colrel(got.cpu().numpy(), ref) < RTOL
        assert not o[1].cpu().numpy()[7].any()

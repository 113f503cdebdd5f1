# -*- coding: utf-8 -*-
"""The launch-count work for small matrices (configs[1]): zero-fills riding on the factor preparation, the gene-tile
split of the row pass, Z += F * R folded into the Gamma update, both M-steps in one launch.  Each fused form against
the separate kernels it replaces.  GPU only."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from oriana_amd import engine
    assert torch.cuda.is_available()
    return engine


def _split_by_default():
    import os
    return os.environ.get('ORIANA_ROW_SPLIT', 'auto') != 'off'


def _two_lane_rows(K):
    """Does the row pass of this K run on the two-lane kernels (the only ones that split single row blocks)?"""
    return 85 <= K <= 100 or 33 <= K <= 64


def _counts(rng, n, m, density):
    X = rng.poisson(3.0, size=(n, m)).astype(np.int64) + 1
    X *= (rng.random((n, m)) < density)
    return X


def test_prep_clears_the_listed_buffers(eng):
    """Every buffer on the list is zero after the launch, whatever its alignment and length; its neighbours are not
    touched; FU / FV equal those of the plain entry."""
    from oriana_amd._lib import call, ptr, stream_ptr
    rng = np.random.default_rng(0)
    n, m, K = 300, 270, 20
    ct = eng.CountTiles.from_dense(_counts(rng, n, m, 0.2), 'cuda')
    ws = eng.ZWorkspace(ct, K)
    lu = torch.randn(n, K, device='cuda')
    lv = torch.randn(m, K, device='cuda')
    eng.factor_prep_pair(ws, lu, lv)
    FU0, FV0 = ws.FU.clone(), ws.FV.clone()
    ws.FU.fill_(7.0); ws.FV.fill_(7.0)
    arena = torch.full((200000,), 3.0, device='cuda')
    spans = [(0, 1), (5, 6), (17, 4096), (5001, 5004), (9997, 70000 + 3), (90001, 90001 + 257), (100000, 100000 + 16),
             (150002, 150002 + 1023)]
    views = [arena[a:b] for a, b in spans]
    eng.factor_prep_pair(ws, lu, lv, clear=views)
    torch.cuda.synchronize()
    assert torch.equal(ws.FU, FU0) and torch.equal(ws.FV, FV0)
    expect = torch.full((200000,), 3.0)
    for a, b in spans:
        expect[a:b] = 0.0
    assert torch.equal(arena.cpu(), expect)
    # integer and float64 buffers, and a second call with the same list (cached struct)
    flags = torch.ones(37, dtype=torch.int32, device='cuda')
    sums = torch.ones(2, 20, dtype=torch.float64, device='cuda')
    for _ in range(2):
        flags.fill_(1); sums.fill_(1.0)
        eng.factor_prep_pair(ws, lu, lv, clear=(flags, None, sums))
        assert int(flags.abs().sum()) == 0 and float(sums.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        eng.factor_prep_pair(ws, lu, lv, clear=[arena[i:i + 1] for i in range(9)])
    # a byte count that is not a multiple of 4 is an argument error of the C entry
    from oriana_amd import _lib
    cl = _lib.OrianaClearList()
    cl.ptr[0] = arena.data_ptr(); cl.bytes[0] = 6
    rc = _lib.load().oriana_factor_prep_pair_clear(ptr(ws.FU), ptr(ws.FV), ptr(lu), ptr(lv), None, None, None, n, m, K,
                                                   ptr(ws.stats), ctypes.byref(cl), stream_ptr())
    assert rc == -1


@pytest.mark.parametrize('K', [5, 20, 50, 64, 100, 128, 200])
def test_row_pass_gene_split_matches_one_group_per_row_block(eng, K):
    """oriana_row_pass_split over 1, 2 and ncb groups per row block: s is the same bit for bit, the slabs of R add up to
    the one-group row sums up to the order of the float additions."""
    from oriana_amd._lib import call, ptr, stream_ptr
    rng = np.random.default_rng(K)
    n, m = 530, 900
    ct = eng.CountTiles.from_dense(_counts(rng, n, m, 0.15), 'cuda')
    assert ct.ncb == 4
    ws = eng.ZWorkspace(ct, K)
    auto = int(ws.row_gene_splits)
    assert 1 <= auto <= ct.ncb
    if _split_by_default():
        assert auto > 1                                           # 3 row blocks: far from filling the chip
    lu = torch.randn(n, K, device='cuda')
    lv = torch.randn(m, K, device='cuda')
    eng.factor_prep_pair(ws, lu, lv)
    st = stream_ptr()
    ws.tile_flag.zero_()
    call('oriana_row_pass', ct.sparse_struct, ptr(ws.FU), ptr(ws.FV), None, ptr(ws.R), ptr(ws.s_cs), None, None,
         ptr(ws.tile_flag), K, st)
    R0, s0 = ws.R.reshape(-1, n, ws.Kp)[0].clone(), ws.s_cs.clone()
    for gs in (1, 2, ct.ncb):
        slabs = torch.full((gs, n, ws.Kp), 7.0, device='cuda')              # (every slab is overwritten: nothing to clear)
        ws.s_cs.zero_()
        call('oriana_row_pass_split', ct.sparse_struct, ptr(ws.FU), ptr(ws.FV), ptr(slabs), ptr(ws.s_cs), ptr(ws.tile_flag),
             K, gs, st)
        torch.cuda.synchronize()
        assert torch.equal(ws.s_cs, s0), gs
        if gs == 1:
            assert torch.equal(slabs[0], R0)
        else:
            scale = R0.abs().max(0, keepdim=True).values.clamp_min(1e-30)
            assert float(((slabs.sum(0) - R0).abs() / (R0.abs() + scale)).max()) < 2e-6, gs
            # the consumer adds the slabs up: oriana_finalize_slabs against oriana_finalize on the one-group sums
            Za = torch.zeros(n, K, device='cuda'); Zb = torch.zeros(n, K, device='cuda')
            call('oriana_finalize', ptr(Za), ptr(ws.FU), ptr(R0), None, ptr(ct.row_perm), n, K, 1, st)
            call('oriana_finalize_slabs', ptr(Zb), ptr(ws.FU), ptr(slabs), gs, ptr(ct.row_perm), n, K, st)
            zs = Za.abs().max(0, keepdim=True).values.clamp_min(1e-30)
            assert float(((Za - Zb).abs() / (Za.abs() + zs)).max()) < 2e-6
    slabs = torch.zeros(ct.ncb + 1, n, ws.Kp, device='cuda')
    rc = __import__('oriana_amd')._lib.load().oriana_row_pass_split(ct.sparse_struct, ptr(ws.FU), ptr(ws.FV), ptr(slabs),
                                                                    ptr(ws.s_cs), ptr(ws.tile_flag), K, ct.ncb + 1, st)
    assert rc == -1


def test_gene_split_rule(eng):
    """1 from 256 row-side work-groups on; otherwise enough groups for two per CU, at most one per gene tile."""
    if not _split_by_default():
        pytest.skip('ORIANA_ROW_SPLIT=off overrides the rule')
    rng = np.random.default_rng(1)
    small = eng.CountTiles.from_dense(_counts(rng, 300, 2000, 0.05), 'cuda')       # 2 row blocks x 8 gene tiles
    assert eng.ZWorkspace(small, 20).row_gene_splits == 8 and eng.ZWorkspace(small, 20).R.shape == (8 * 300, 20)
    one_tile = eng.CountTiles.from_dense(_counts(rng, 300, 200, 0.05), 'cuda')
    assert eng.ZWorkspace(one_tile, 20).row_gene_splits == 1
    import scipy.sparse as sp
    tall = eng.CountTiles.from_scipy(sp.random(70000, 600, density=0.01, format='csr', random_state=2, dtype=np.float32), 'cuda')
    assert tall.nrb >= 256 and eng.ZWorkspace(tall, 20).row_gene_splits == 1
    # [r4] the one-group-per-CU kernels (33 <= Kp <= 64, 85 <= K <= 100) split the row blocks of a partly filled LAST round of
    # the chip: 391 row blocks = 256 whole + 135 in five gene ranges (3 / 5 of a round); 489 = 256 + 233: two rounds anyway
    import os
    if True:
        c3 = eng.CountTiles.from_scipy(sp.random(100000, 1200, density=0.004, format='csr', random_state=3, dtype=np.float32), 'cuda')
        assert c3.nrb == 391 and c3.ncb == 5
        for K in (50, 100):                                                      # the 135 row blocks of the second round in 5 ranges
            w = eng.ZWorkspace(c3, K)
            assert (w.row_split.nfull, w.row_split.parts, w.row_slab_row0) == (256, 5, 65536)
            assert w.R.shape[0] == 100000 + 4 * (100000 - 65536)          # slab 0 for every row, slabs 1-4 for the split row blocks only
            assert list(w.row_split.edge[:6]) == [0, 1, 2, 3, 4, 5]
        assert eng.ZWorkspace(c3, 20).row_gene_splits == 1                       # (narrow kernels: several groups per CU)
        c8 = eng.CountTiles.from_scipy(sp.random(125000, 1200, density=0.004, format='csr', random_state=4, dtype=np.float32), 'cuda')
        assert c8.nrb == 489 and eng.ZWorkspace(c8, 100).row_gene_splits == 1    # (233 row blocks in the last round: nothing to gain)


@pytest.mark.parametrize('K', [20, 50, 64, 100])
@pytest.mark.parametrize('nest', ['zi', 'zi_quirk', 'sparse', 'sparse_zi', 'weights'])
def test_zq_with_gene_splits_matches_one_group_per_row_block(eng, K, nest):
    """[r4] engine.zq hands every variant of the row pass (plain, D_hat[i, k] quirk, per-entry weights, fused and unfused sparse
    forms) the gene-tile split of its workspace: against a workspace forced to one group per row block, float32 rounding apart."""
    rng = np.random.default_rng(K + len(nest))
    n, m = 530, 900
    X = _counts(rng, n, m, 0.15)
    lu = torch.from_numpy(rng.normal(size=(n, K)).astype(np.float32)).cuda()
    lv = torch.from_numpy((rng.normal(size=(m, K)) - 0.5).astype(np.float32)).cuda()
    ps = rng.random((m, K))
    kw = {}
    if nest.startswith('sparse'):
        kw['S_tilde'] = torch.from_numpy((ps > 0.3).astype(np.float32)).cuda()
        kw['S_hat'] = torch.from_numpy(ps.astype(np.float32)).cuda()
    if nest in ('zi_quirk', 'sparse_zi'):
        kw['dq'] = torch.rand(n, K, device='cuda')
    side = None
    if nest == 'weights':
        side = torch.from_numpy(rng.random((n, m)).astype(np.float32)).cuda()
    ct = eng.CountTiles.from_dense(X, 'cuda', side=side)
    if nest == 'weights':
        kw['w_nz'] = ct.side_nz

    def run(splits):
        ws = eng.ZWorkspace(ct, K, need_sw=nest == 'weights')
        if splits is not None:
            ws.set_row_split(*splits)
        o = [torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')]
        eng.zq(ws, o[0], o[1], o[2] if nest != 'weights' else None, lu, lv, **kw)
        torch.cuda.synchronize()
        return o[:2] + ([o[2]] if nest != 'weights' else [])
    assert ct.nrb == 3 and ct.ncb == 4
    ref = run((3, 1, (0, 4)))
    # every row block in 2 / 4 even ranges; the last two row blocks in 3 uneven ranges, the last one in 2 (the first in one piece)
    for splits in ((0, 2), (0, 4), (1, 3, (0, 1, 2, 4)), (2, 2, (0, 3, 4))):
        if splits[0] > 0 and not _two_lane_rows(K):
            continue                                               # (the other kernels split whole grids only)
        got = run(splits)
        for a, b in zip(got, ref):
            sc = b.abs().max(0, keepdim=True).values.clamp_min(1e-30)
            assert float(((a - b).abs() / (b.abs() + sc)).max()) < 3e-6, (splits, nest)


@pytest.mark.parametrize('K', [50, 100])
@pytest.mark.parametrize('nest', ['gap', 'zi_quirk', 'sparse'])
def test_hybrid_layout_with_a_split_last_round(eng, K, nest):
    """[r4] Hybrid layout: the dense row kernel splits the same 256-cell blocks as the sliced row pass and adds its parts into the
    same slabs of R (oriana_dense_row_pass_tail) -- against the unsplit workspace of the same matrix."""
    rng = np.random.default_rng(K)
    n, m = 700, 900
    dens = np.clip(rng.beta(1.0, 2.0, size=m), 0.02, 1.0)
    dens[:160] = np.linspace(1.0, 0.5, 160)
    X = (rng.poisson(3.0, size=(n, m)) + 1) * (rng.random((n, m)) < dens[None, :])
    ct = eng.CountTiles.from_dense(X.astype(np.float32), 'cuda', dense_density=0.45)
    assert ct.gd >= 128 and ct.ms > 0 and ct.nrb == 3
    lu = torch.from_numpy(rng.normal(size=(n, K)).astype(np.float32)).cuda()
    lv = torch.from_numpy((rng.normal(size=(m, K)) - 0.5).astype(np.float32)).cuda()
    ps = rng.random((m, K))
    kw = {}
    if nest == 'sparse':
        kw = dict(S_tilde=torch.from_numpy((ps > 0.3).astype(np.float32)).cuda(), S_hat=torch.from_numpy(ps.astype(np.float32)).cuda())
    if nest == 'zi_quirk':
        kw = dict(dq=torch.rand(n, K, device='cuda'))

    def run(split):
        ws = eng.ZWorkspace(ct, K)
        ws.dn_gene_splits = 1
        ws.set_row_split(*split)
        o = [torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda'), torch.empty(m, K, device='cuda')]
        if nest == 'gap':
            eng.zq_gap(ws, o[0], o[1], lu, lv)
            o = o[:2]
        else:
            eng.zq(ws, o[0], o[1], o[2], lu, lv, **kw)
        torch.cuda.synchronize()
        tail = ws.dense_tail(ws.row_gene_splits)
        return o, tail
    ref, t0 = run((ct.nrb, 1, (0, ct.ncb)))
    assert t0 == (0, 1)
    import os
    for split in ((1, 2, None), (2, 3, None), (1, 2, (0, 1, ct.ncb))):
        if split[1] > ct.ncb or not _two_lane_rows(K):
            continue
        got, tail = run(split)
        # (what a call with that many slabs hands the dense kernel; the unfused sparse form, K > 64, has one slab and no dense split)
        assert tail == (split[0], split[1])
        for a, b in zip(got, ref):
            sc = b.abs().max(0, keepdim=True).values.clamp_min(1e-30)
            assert float(((a - b).abs() / (b.abs() + sc)).max()) < 3e-6, (split, nest)


@pytest.mark.parametrize('r,K,perm,nslab', [(1000, 20, True, 1), (257, 100, False, 1), (3, 5, True, 3), (5000, 7, True, 4),
                                            (40000, 20, False, 1), (40000, 100, True, 1), (33000, 85, False, 3), (33001, 127, True, 2)])
def test_gamma_update_with_folded_finalize(eng, r, K, perm, nslab):
    """oriana_gamma_update_finalize == oriana_finalize_slabs followed by oriana_gamma_update, bit for bit (both group
    sizes of the update: 1024 threads up to 32768 rows, 256 beyond)."""
    from oriana_amd._lib import call, ptr, stream_ptr
    g = torch.Generator(device='cuda').manual_seed(r + K)
    Kp = eng.kpad(K)
    dev = 'cuda'
    F = torch.rand(r, Kp, device=dev, generator=g)
    R = torch.rand(nslab, r, Kp, device=dev, generator=g) * 50
    Zfix = torch.rand(r, K, device=dev, generator=g) * (torch.rand(r, K, device=dev, generator=g) < 0.05)
    idx = torch.randperm(r, device=dev, generator=g).to(torch.int32) if perm else None
    p1 = torch.rand(K, dtype=torch.float64, device=dev, generator=g) + 0.5
    p2 = torch.rand(K, dtype=torch.float64, device=dev, generator=g) + 0.5
    rate = torch.rand(K, dtype=torch.float64, device=dev, generator=g) * 100
    st = stream_ptr()

    def outs():
        return (torch.empty(r, K, dtype=torch.float64, device=dev), torch.empty(r, K, dtype=torch.float64, device=dev),
                torch.empty(r, K, dtype=torch.float64, device=dev), torch.empty(r, K, dtype=torch.float32, device=dev),
                torch.zeros(2, K, dtype=torch.float64, device=dev))
    a1, a2, E, El, sums = outs()
    Z = Zfix.clone()
    call('oriana_finalize_slabs', ptr(Z), ptr(F), ptr(R), nslab, ptr(idx), r, K, st)
    call('oriana_gamma_update', ptr(a1), ptr(a2), ptr(E), ptr(El), ptr(sums[0]), ptr(sums[1]), ptr(p1), ptr(p2), ptr(Z), None,
         ptr(rate), None, None, r, K, st)
    b1, b2, E2, El2, sums2 = outs()
    Z2 = Zfix.clone()
    call('oriana_gamma_update_finalize', ptr(b1), ptr(b2), ptr(E2), ptr(El2), ptr(sums2[0]), ptr(sums2[1]), ptr(p1), ptr(p2),
         ptr(Z2), ptr(F), ptr(R), nslab, ptr(idx), ptr(rate), r, K, st)
    torch.cuda.synchronize()
    assert torch.equal(Z2, Z)
    for x, y in ((a1, b1), (a2, b2), (E, E2), (El, El2)):
        assert torch.equal(x, y)
    # (the column sums are float64 atomics over the same per-block partial sums in a different block order)
    assert torch.allclose(sums, sums2, rtol=1e-12, atol=0)


@pytest.mark.parametrize('K', [1, 20, 100, 200])
def test_mstep_pair_matches_two_launches(eng, K):
    from oriana_amd._lib import call, ptr, stream_ptr
    g = torch.Generator(device='cuda').manual_seed(K)
    dev = 'cuda'

    def vec(lo, hi):
        return torch.rand(K, dtype=torch.float64, device=dev, generator=g) * (hi - lo) + lo
    cu, cv = 12345.0, 777.0
    sEu, sLu, sEv, sLv = vec(1e3, 1e5), vec(-3e4, 3e4), vec(1e2, 1e4), vec(-2e3, 2e3)
    p = [vec(0.5, 3.0) for _ in range(4)]
    q = [t.clone() for t in p]
    st = stream_ptr()
    call('oriana_mstep_gamma', ptr(p[0]), ptr(p[1]), ptr(sEu), ptr(sLu), cu, K, st)
    call('oriana_mstep_gamma', ptr(p[2]), ptr(p[3]), ptr(sEv), ptr(sLv), cv, K, st)
    accv = torch.stack([sEv, sLv]).contiguous()
    keep = torch.zeros(2, K, dtype=torch.float64, device=dev)
    call('oriana_mstep_gamma_pair', ptr(q[0]), ptr(q[1]), ptr(sEu), ptr(sLu), cu, ptr(q[2]), ptr(q[3]), ptr(accv[0]),
         ptr(accv[1]), cv, ptr(keep), K, st)
    torch.cuda.synchronize()
    for x, y in zip(p, q):
        assert torch.equal(x, y)
    assert torch.equal(keep, accv)


def test_sweep_is_the_same_with_and_without_the_small_matrix_forms(eng, monkeypatch):
    """pCMF sweeps on a configs[1]-shaped problem (scaled down): gene split on (default) and off give the same state to
    float-addition order; an E-step without the M-step before it still sees sum_j V_hat."""
    from oriana_amd.models import GaP
    from oracle import cavi_oracle as co
    rng = np.random.default_rng(5)
    n, m, K = 700, 1100, 20
    X = _counts(rng, n, m, 0.1)
    a1 = rng.gamma(1.0, size=(n, K)); b1 = rng.gamma(1.0, size=(m, K))
    monkeypatch.delenv('ORIANA_ROW_SPLIT', raising=False)
    A = GaP(X, k=K, init=(a1, b1), device='cuda')
    assert A._ws.row_gene_splits > 1 or not _split_by_default()
    monkeypatch.setenv('ORIANA_ROW_SPLIT', 'off')
    B = GaP(X, k=K, init=(a1, b1), device='cuda')
    assert B._ws.row_gene_splits == 1
    ref = co.OracleGaP(X, K, a1, b1)
    for _ in range(3):
        A.step(); B.step(); ref.step()
    sa, sb, sr = A.state(), B.state(), ref.state()
    for key in ('a1', 'a2', 'b1', 'b2', 'alpha1', 'alpha2', 'beta1', 'beta2'):
        np.testing.assert_allclose(sa[key], sb[key], rtol=2e-6, err_msg=key)
        np.testing.assert_allclose(sa[key], sr[key], rtol=1e-5, err_msg=key)
    # two E-steps in a row (no M-step between them): the reference recomputes V_hat.sum(axis=0) each time (gap.py:98)
    A.update_variational_parameters(); A.update_variational_parameters(); A.update_prior_hyper_parameters()
    ref.update_variational_parameters(); ref.update_variational_parameters(); ref.update_prior_hyper_parameters()
    sa, sr = A.state(), ref.state()
    for key in ('a1', 'a2', 'b1', 'b2', 'alpha1', 'alpha2', 'beta1', 'beta2'):
        np.testing.assert_allclose(sa[key], sr[key], rtol=1e-5, err_msg=key)

# -*- coding: utf-8 -*-
"""[r5] The cell side of the factor preparation folded into the Gamma update that produces E[log U]
(oriana_gamma_update_prep / oriana_gamma_update_finalize_prep + oriana_factor_prep_pair_fused) against the separate
launches it replaces (k_row_stats + k_factor_prep_pair over E[log U]): same FU bit for bit, same statistics up to the
summation order, same sweeps.  Reference: gap.py:96-110 (the update), gap.py:72-80 (what FU feeds).  GPU only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from oriana_amd import engine
    assert torch.cuda.is_available()
    return engine


def _counts(rng, n, m, density):
    X = rng.poisson(3.0, size=(n, m)).astype(np.int64) + 1
    X *= (rng.random((n, m)) < density)
    return X


SHAPES = [(300, 20), (1000, 100), (777, 50), (515, 64), (260, 5), (333, 7), (4100, 36), (70000, 100), (129, 256), (200, 128)]


@pytest.mark.parametrize('r,K', SHAPES)
@pytest.mark.parametrize('form', ['plain', 'finalize'])
def test_update_with_prep_outputs_matches_separate_preparation(eng, r, K, form):
    from oriana_amd import _lib
    from oriana_amd._lib import call, ptr, stream_ptr
    lib = _lib.load()
    nblk = int(lib.oriana_gamma_update_prep_blocks(r, K))
    assert nblk > 0
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(r * 131 + K)
    Kp = eng.kpad(K)
    st = stream_ptr()
    p1 = torch.rand(K, dtype=torch.float64, device=dev, generator=g) + 0.5
    p2 = torch.rand(K, dtype=torch.float64, device=dev, generator=g) + 0.5
    rate = torch.rand(K, dtype=torch.float64, device=dev, generator=g) * 50
    Z = torch.rand(r, K, device=dev, generator=g) * 30
    # a row whose logs lie far from the others (rejected by the centred test), a zero row, a NaN count sum (clamped to 1e-15)
    Z[3] = 1e30
    Z[7] = 0.0
    if r > 50:
        Z[41, K // 2] = float('nan')
    outs = {}
    for mode in ('sep', 'prep'):
        a1, a2, E = (torch.empty(r, K, dtype=torch.float64, device=dev) for _ in range(3))
        El = torch.empty(r, K, device=dev)
        sums = torch.zeros(2, K, dtype=torch.float64, device=dev)
        FUn = torch.zeros(r, Kp, device=dev)
        mu = torch.zeros(r, device=dev)
        up = torch.zeros(4 * nblk, device=dev)
        prep = (ptr(FUn), ptr(mu), ptr(up)) if mode == 'prep' else (None, None, None)
        if form == 'plain':
            call('oriana_gamma_update_prep', ptr(a1), ptr(a2), ptr(E), ptr(El), ptr(sums[0]), ptr(sums[1]), ptr(p1), ptr(p2),
                 ptr(Z), None, ptr(rate), None, None, r, K, prep[0], prep[1], prep[2], st)
        else:
            F = torch.ones(r, Kp, device=dev)
            # slab 0: every row; slab 1: the rows from r // 2 on only (stride r - r // 2 rows, oriana_finalize_slabs_from)
            row0 = r // 2
            R = torch.zeros(r + (r - row0), Kp, device=dev)
            R[:r, :K] = Z * 0.75
            R[:row0, :K] = Z[:row0]
            R[r:, :K] = Z[row0:] * 0.25
            Zf = torch.zeros(r, K, device=dev)
            call('oriana_gamma_update_finalize_prep', ptr(a1), ptr(a2), ptr(E), ptr(El), ptr(sums[0]), ptr(sums[1]), ptr(p1), ptr(p2),
                 ptr(Zf), ptr(F), ptr(R), 2, r // 2, None, ptr(rate), r, K, prep[0], prep[1], prep[2], st)
        outs[mode] = dict(a1=a1, a2=a2, E=E, El=El, sums=sums, FUn=FUn, mu=mu, up=up)
    torch.cuda.synchronize()
    for k in ('a1', 'a2', 'E', 'El'):
        assert torch.equal(outs['sep'][k].nan_to_num(nan=-7.0), outs['prep'][k].nan_to_num(nan=-7.0)), k
    El = outs['prep']['El']
    # the update itself against float64 torch (gap.py:97-100, gamma.py:37-61)
    a1_ref = torch.clamp(torch.nan_to_num(p1 + Z.double(), nan=0.0), min=1e-15)
    ok = torch.isfinite(a1_ref)
    assert torch.allclose(outs['prep']['a1'][ok], a1_ref[ok], rtol=1e-6 if form == 'finalize' else 0, atol=0)
    el_ref = (torch.digamma(a1_ref.float().double()).float() - torch.log((p2 + rate).float())[None, :])
    fin = torch.isfinite(el_ref) & torch.isfinite(El)
    assert float((El[fin] - el_ref[fin]).abs().max()) <= 2e-6 * float(el_ref[fin].abs().max())
    # the separate preparation over the same E[log U]
    m = 64
    lv = torch.randn(m, K, device=dev, generator=g)
    scratch_a = torch.zeros(int(lib.oriana_prep_scratch_bytes()) // 4, device=dev)
    scratch_b = torch.zeros_like(scratch_a)
    FUa, FVa = torch.zeros(r, Kp, device=dev), torch.zeros(m, Kp, device=dev)
    FVb = torch.zeros(m, Kp, device=dev)
    call('oriana_factor_prep_pair', ptr(FUa), ptr(FVa), ptr(El), ptr(lv), None, None, None, r, m, K, ptr(scratch_a), st)
    FUb = outs['prep']['FUn']
    call('oriana_factor_prep_pair_fused', ptr(FUb), ptr(outs['prep']['mu']), ptr(outs['prep']['up']), nblk, ptr(FVb), ptr(lv), None, None,
         r, m, K, ptr(scratch_b), None, st)
    torch.cuda.synchronize()
    sa, sb = scratch_a[:10].cpu().numpy(), scratch_b[:10].cpu().numpy()
    assert sa[2] == sb[2] and sa[5] == sb[5]                              # the counts of the rows that entered the statistics
    np.testing.assert_allclose(sb[[0, 1, 3, 4]], sa[[0, 1, 3, 4]], rtol=2e-5)
    assert sa[8] == sb[8] and sa[9] == sb[9] and sa[7] == sb[7]          # smallest row maxima, the den threshold
    assert torch.equal(FVa, FVb)
    assert torch.equal(FUa, FUb), int((FUa != FUb).any(1).sum())
    # the rejected row holds the fill constant (1e-30)
    assert float(FUb[3, 0]) == pytest.approx(1e-30)


def test_odd_wide_k_keeps_the_separate_preparation(eng):
    from oriana_amd import _lib
    lib = _lib.load()
    assert int(lib.oriana_gamma_update_prep_blocks(1000, 101)) == 0
    assert int(lib.oriana_gamma_update_prep_blocks(1000, 300)) == 0
    assert int(lib.oriana_gamma_update_prep_blocks(1000, 63)) > 0


@pytest.mark.parametrize('model_name,K', [('GaP', 20), ('GaP', 100), ('ZIGaP', 50), ('SparseGaP', 64), ('SparseZIGaP', 36)])
def test_models_with_and_without_the_fused_preparation(eng, model_name, K, monkeypatch):
    """Four free-running sweeps with the cell side prepared by the Gamma update against the same model with the separate
    launches (ORIANA_FUSED_PREP=off): FU is the same bit for bit, so the states agree to the order of the float atomics."""
    import oriana_amd
    from oriana_amd import models
    rng = np.random.default_rng(5)
    n, m = 1300, 420
    X = _counts(rng, n, m, 0.15)
    a1 = rng.gamma(1.0, size=(n, K)) + 0.05
    b1 = rng.gamma(1.0, size=(m, K)) + 0.05
    cls = getattr(models, model_name)
    states = {}
    eng.set_deterministic(True)
    try:
        for mode in ('force', 'off'):                  # (force: also below the 2^20 elements from which 'on' takes it)
            monkeypatch.setenv('ORIANA_FUSED_PREP', mode)
            mdl = cls(X, k=K, init=(a1, b1), dense_density=None)
            assert (mdl._ws.prep_blocks > 0) == (mode == 'force')
            for _ in range(4):
                mdl.step()
            if mode == 'force':
                assert mdl._ws.fu_pending and mdl._ws.FU_alt is not None
            states[mode] = mdl.state()
    finally:
        eng.set_deterministic(False)
    for k, v in states['force'].items():
        ref = states['off'][k]
        if v.dtype.kind == 'f':
            scale = np.abs(ref).max(axis=0, keepdims=True) if ref.ndim == 2 else np.abs(ref).max()
            err = np.abs(v - ref) / (np.abs(ref) + scale + 1e-300)
            assert float(np.nanmax(err)) <= 2e-6, (k, float(np.nanmax(err)))


def test_a_foreign_log_matrix_does_not_take_the_pending_preparation(eng, monkeypatch):
    """engine.zq_gap on the model's workspace with ANOTHER E[log U] (bench.py's parity slab, tests) must prepare from that
    matrix, not from what the last sweep left."""
    from oriana_amd import models
    rng = np.random.default_rng(6)
    n, m, K = 600, 300, 20
    X = _counts(rng, n, m, 0.2)
    monkeypatch.setenv('ORIANA_FUSED_PREP', 'force')
    mdl = models.GaP(X, k=K, init=(rng.gamma(1.0, size=(n, K)) + 0.1, rng.gamma(1.0, size=(m, K)) + 0.1), dense_density=None)
    mdl.step()
    ws = mdl._ws
    assert ws.fu_pending
    lu = torch.randn(n, K, device='cuda')
    lv = torch.randn(m, K, device='cuda')
    Zi, Zj = torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda')
    eng.zq_gap(ws, Zi, Zj, lu, lv)
    ct2 = eng.CountTiles.from_dense(X, 'cuda')
    ws2 = eng.ZWorkspace(ct2, K)
    Zi2, Zj2 = torch.empty(n, K, device='cuda'), torch.empty(m, K, device='cuda')
    eng.zq_gap(ws2, Zi2, Zj2, lu, lv)
    torch.cuda.synchronize()
    assert torch.allclose(Zi, Zi2, rtol=1e-5, atol=1e-6) and torch.allclose(Zj, Zj2, rtol=1e-5, atol=1e-5)

# -*- coding: utf-8 -*-
"""Shared helpers for the parity tests (tolerances follow SURVEY.md section 7, hard part 4)."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

PARAM_KEYS = ['alpha1', 'alpha2', 'beta1', 'beta2', 'a1', 'a2', 'b1', 'b2', 'pi_d', 'p_d', 'pi_s', 'p_s']
EXPECT_KEYS = ['U_hat', 'V_hat', 'log_U_hat', 'log_V_hat', 'S_hat']

RTOL = 1e-5   # BASELINE.json north_star: "within 1e-5 relative on the variational parameters"
# Achieved errors of the HIP path against the reference's golden states, all models / shapes / starts
# (profiles/r02_parity_errors.json, tools/parity_report.py): Gamma parameters and expectations <= 1.6e-6 in this
# metric and <= 3.3e-6 STRICTLY element-wise relative, i.e. RTOL has a 3x margin even without the column term.
# The dropout posterior is a probability: its error is judged absolutely -- measured 1.2e-7 on p_d and 7e-10 on
# pi_d against the reference (round 1 allowed 1e-4); the bounds below leave room for the larger Lambda of the
# multi-tile / sharded cases (|d p_d| <= 0.25 |d Lambda|, Lambda carries the 1e-6 relative error of U_hat, V_hat).
KEY_ATOL = {'p_d': 2e-6, 'pi_d': 1e-7}
# The sparsity posterior p_s = sigmoid(logit(pi_s) - t), t a float32 difference of two sums of magnitude 1e4..1e6,
# is judged against the EXACT value of the same sweep wherever a test can compute it (exact_twin below: HIP within
# 7.1e-6, the reference's own arithmetic within 2.0e-6).  Where it cannot (the NMF-start goldens, whose float32 loop
# underflows where float64 does not), the bound against the reference's state is 2 x the worst error achieved over all
# goldens and sweeps (profiles/r03_parity_errors.json: p_s, S_hat 5.8e-6 in the column metric; pi_s 5.4e-7, i.e. inside
# RTOL without an entry here).  Round 3 allowed 1e-4.
KEY_RTOL = {'p_s': 1.2e-5, 'S_hat': 1.2e-5}

def golden_files(pattern='*_*.npz'):
    skip_metrics = not pattern.startswith('metrics_')
    return sorted(f for f in glob.glob(os.path.join(GOLDEN, pattern))
                  if not f.endswith('tables.npz') and not os.path.basename(f).startswith('generator_')
                  and not f.endswith('_xcheck.npz')
                  and not (skip_metrics and os.path.basename(f).startswith('metrics_')))


def xcheck_files():
    """The patched-SparseGaP fixtures cross-checked by the reference's unpatched SparseZIGaP class with D_hat == 1
    (tests/golden/make_golden.py xcheck; SURVEY 8a policy ii)."""
    return sorted(glob.glob(os.path.join(GOLDEN, 'sparsegap_*_xcheck.npz')))


def load_golden(path):
    g = np.load(path)
    return {k: g[k] for k in g.files}


def state_of(g, tag):
    pre = tag + '/'
    return {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}


def err_colrel(got, ref):
    """max |got-ref| / (|ref| + colmax|ref|): the per-column relative error of SURVEY 7.4."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    if ref.ndim == 2:
        cm = np.max(np.abs(ref), axis=0, keepdims=True)
    else:
        cm = np.max(np.abs(ref))
    den = np.abs(ref) + cm + 1e-300
    bad = ~np.isfinite(got) | ~np.isfinite(ref)
    if bad.any():
        same = (got == ref) | (np.isnan(got) & np.isnan(ref))
        if not same[bad].all():
            return np.inf
    d = np.where(bad, 0.0, np.abs(got - ref) / den)
    return float(np.max(d))


def exact_twin(oracle_model):
    """A copy of an oracle model (BEFORE its step) whose loop nest runs in float64 (cavi_oracle.zq_exact): the yardstick
    the reference's float32 evaluation and the HIP path are both measured against for the sparsity posterior.

    p_s = sigmoid(logit(pi_s) - t), t = -Zlog + c * Vprime (sparse_gap.py:135-137), is a difference of two sums of
    magnitude 1e3..1e6 that leaves t = O(10): every float32 evaluation of the loop nest -- the reference's own included --
    moves it.  Measured on the goldens (profiles/r03_parity_errors.json): the reference is up to 2.0e-6 away from the
    exact p_s, the HIP path up to 7.1e-6 (1.4e-4 in round 2, before the log sums were centred)."""
    import copy
    E = copy.deepcopy(oracle_model)
    E.exact = True
    return E


PS_FLOOR = {'p_s': 1.4e-5, 'S_hat': 1.4e-5, 'pi_s': 2e-6}   # absolute, against the EXACT value: 2 x the measured worst (7.1e-6 / 1.0e-6)
PS_FACTOR = 8.0                                            # ... or this many times the reference's own distance from exact (<= 2.0e-6)


def assert_state_close(got, ref, rtol=RTOL, keys=None, what='', exact=None):
    """|got - ref| <= rtol*|ref| + rtol*colmax|ref| on every key, plus identical clamp
    patterns (entries sitting exactly on the 1e-15 floor / the 1-1e-10 ceiling).  `exact`: the state of the exact twin
    of the same sweep -- the sparsity posterior is then judged against IT: the HIP value may be no further from exact
    than PS_FLOOR, or PS_FACTOR times the distance of `ref` (the reference's float32 arithmetic) from exact."""
    keys = keys or [k for k in PARAM_KEYS + EXPECT_KEYS if k in ref]
    for k in keys:
        if k not in ref or k not in got:
            continue
        if exact is not None and k in PS_FLOOR:
            d_hip = float(np.abs(np.asarray(got[k], dtype=np.float64) - np.asarray(exact[k], dtype=np.float64)).max())
            d_ref = float(np.abs(np.asarray(ref[k], dtype=np.float64) - np.asarray(exact[k], dtype=np.float64)).max())
            assert d_hip <= max(PS_FLOOR[k], PS_FACTOR * d_ref), \
                '%s %s: HIP is %.3e from exact, the reference arithmetic %.3e' % (what, k, d_hip, d_ref)
            continue
        if k in KEY_ATOL:
            e = float(np.max(np.abs(np.asarray(got[k], dtype=np.float64) - np.asarray(ref[k], dtype=np.float64)))) if np.size(ref[k]) else 0.0
            assert e <= KEY_ATOL[k], '%s %s: abs err %.3e > %.1e' % (what, k, e, KEY_ATOL[k])
        else:
            e = err_colrel(got[k], ref[k])
            tol = max(rtol, KEY_RTOL.get(k, 0.0))
            assert e <= tol, '%s %s: err %.3e > %.1e' % (what, k, e, tol)
        if k in ('a1', 'a2', 'b1', 'b2', 'alpha1', 'alpha2', 'beta1', 'beta2'):
            assert np.array_equal(np.asarray(got[k]) == 1e-15, np.asarray(ref[k]) == 1e-15), \
                '%s %s: 1e-15 clamp pattern differs' % (what, k)
        if k == 'p_d':
            assert np.array_equal(np.asarray(got[k]) == 1. - 1e-10, np.asarray(ref[k]) == 1. - 1e-10), \
                '%s p_d: (1 - 1e-10) mask differs' % what


def dropout_sweep(D, U, V, pi, mask, cs, Vn, DV, scratch, arithmetic, n, m, K):
    """oriana_dropout_sweep_fused_tiles with the per-lane flags built from `mask` (oriana_nzmask_f32 layout; None: no mask,
    which is the round-2 entry oriana_dropout_sweep_fused)."""
    import torch
    from oriana_amd import _lib
    from oriana_amd._lib import call, ptr, stream_ptr
    tiles = None
    if mask is not None:
        tiles = torch.zeros(max(int(_lib.load().oriana_nzmask_tiles_words(n, m)), 4), dtype=torch.int32, device=mask.device)
        call('oriana_nzmask_tiles', ptr(tiles), ptr(mask), n, m, stream_ptr())
    call('oriana_dropout_sweep_fused_tiles', ptr(D), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(tiles), ptr(cs), ptr(Vn), ptr(DV),
         ptr(scratch), arithmetic, n, m, K, stream_ptr())
    return tiles

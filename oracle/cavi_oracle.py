# -*- coding: utf-8 -*-
"""oracle/cavi_oracle.py -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).

CPU restatement (NumPy + SciPy special functions + the C loop nests of
``oracle/zq_kernels.c``) of the reference's CAVI hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Parity status: PINNED -- checked in ``tests/test_oracle.py`` against golden vectors
captured from the reference itself (imported in the build container with two shims,
see ``tests/golden/make_golden.py``) and against the known answers of the reference's
own unit tests (``test/test.py:13-79``).

Every function cites the reference lines (relative to the reference repo root) it
restates.  State is a plain dict of NumPy arrays using the reference's attribute names.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.special

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EPS = 1e-15  # clamp floor used everywhere in the reference (e.g. gap.py:99)


# ----------------------------------------------------------------------------------------
# C loop nests
# ----------------------------------------------------------------------------------------
def build(force=False):
    """Compile oracle/zq_kernels.c -> oracle/liboracle.so (gcc, -O2, no fast-math)."""
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'zq_kernels.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'liboracle.so'])
    return so


def build_omp(force=False):
    """Compile oracle/zq_kernels_omp.c -> oracle/liboracle_omp.so (the labelled all-cores variant of bench.py)."""
    so = os.path.join(_HERE, 'liboracle_omp.so')
    src = os.path.join(_HERE, 'zq_kernels_omp.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'liboracle_omp.so'])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        i64 = ctypes.c_int64
        _LIB.zq_gap.argtypes = [fp] * 5 + [i64] * 3
        _LIB.zq_zigap.argtypes = [fp] * 7 + [i64] * 3 + [ctypes.c_int]
        _LIB.zq_sparse_gap.argtypes = [fp] * 8 + [i64] * 3
        _LIB.zq_sparse_zigap.argtypes = [fp] * 9 + [i64] * 3
        _LIB.zq_gap_nz.argtypes = [fp] * 5 + [i64] * 3
        _LIB.zq_sparse_gap_nz.argtypes = [fp] * 8 + [i64] * 3
        for f in (_LIB.zq_gap, _LIB.zq_zigap, _LIB.zq_sparse_gap, _LIB.zq_sparse_zigap, _LIB.zq_gap_nz, _LIB.zq_sparse_gap_nz):
            f.restype = ctypes.c_int
    return _LIB


_LIB_OMP = None


def zq_gap_omp(Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X, nthreads):
    """gap.py:67-80 with OpenMP over the cells (oracle/zq_kernels_omp.c): NOT the reference's behaviour (one
    thread, gap.py:67) -- a second, labelled CPU figure for bench.py.  Raises if the OpenMP build is unavailable."""
    global _LIB_OMP
    if _LIB_OMP is None:
        _LIB_OMP = ctypes.CDLL(build_omp())
        fp = ctypes.POINTER(ctypes.c_float)
        _LIB_OMP.zq_gap_omp.argtypes = [fp] * 5 + [ctypes.c_int64] * 3 + [ctypes.c_int]
        _LIB_OMP.zq_gap_omp.restype = ctypes.c_int
    n, K = log_U_hat.shape
    p = log_V_hat.shape[0]
    a = [_f32c(Z_hat_i, (n, K)), _f32c(Z_hat_j, (p, K)), _f32c(log_U_hat), _f32c(log_V_hat, (p, K)), _f32c(X, (n, p))]
    rc = _LIB_OMP.zq_gap_omp(*[_p(x) for x in a], n, p, K, int(nthreads))
    assert rc == 0, rc


def _f32c(a, shape=None):
    """The reference kernels are declared f4[:, :] C-contiguous (gap.py:67)."""
    a = np.asarray(a)
    if a.dtype != np.float32 or a.ndim != 2:
        raise TypeError('expected a 2-D float32 array, got %s ndim=%d' % (a.dtype, a.ndim))
    if not a.flags['C_CONTIGUOUS']:
        raise TypeError('expected a C-contiguous array')
    if shape is not None and a.shape != shape:
        raise ValueError('shape mismatch: %s vs %s' % (a.shape, shape))
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def zq_gap(Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X):
    """gap.py:67-80 -- outputs first, zero-filled by the callee, returns None."""
    n, K = log_U_hat.shape
    p = log_V_hat.shape[0]
    a = [_f32c(Z_hat_i, (n, K)), _f32c(Z_hat_j, (p, K)), _f32c(log_U_hat), _f32c(log_V_hat, (p, K)),
         _f32c(X, (n, p))]
    rc = _lib().zq_gap(*[_p(x) for x in a], n, p, K)
    assert rc == 0


def zq_gap_nz(Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X):
    """gap.py:67-80 with the zero counts skipped (oracle/zq_kernels.c: bit-identical to zq_gap while every exp(lu + lv) is
    finite; for oracle sweeps at sizes where the full nest spends minutes on zeros)."""
    n, K = log_U_hat.shape
    p = log_V_hat.shape[0]
    a = [_f32c(Z_hat_i, (n, K)), _f32c(Z_hat_j, (p, K)), _f32c(log_U_hat), _f32c(log_V_hat, (p, K)),
         _f32c(X, (n, p))]
    rc = _lib().zq_gap_nz(*[_p(x) for x in a], n, p, K)
    assert rc == 0


def zq_sparse_gap_nz(SZ_hat_i, Z_hat_j, Z_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, X):
    """sparse_gap.py:81-97 with the zero counts skipped (see zq_gap_nz)."""
    n, K = log_U_hat.shape
    p = log_V_hat.shape[0]
    a = [_f32c(SZ_hat_i, (n, K)), _f32c(Z_hat_j, (p, K)), _f32c(Z_exp_logsum_hat, (p, K)),
         _f32c(log_U_hat), _f32c(log_V_hat, (p, K)), _f32c(S_tilde, (p, K)), _f32c(S_hat, (p, K)),
         _f32c(X, (n, p))]
    rc = _lib().zq_sparse_gap_nz(*[_p(x) for x in a], n, p, K)
    assert rc == 0


def zq_zigap(DZ_hat_i, DZ_hat_j, DZ_exp_logsum_hat, log_U_hat, log_V_hat, D_hat, X, quirk=True):
    """zigap.py:79-95 (quirk=True keeps the D_hat[i, k] index of zigap.py:94)."""
    n, K = log_U_hat.shape
    p = log_V_hat.shape[0]
    a = [_f32c(DZ_hat_i, (n, K)), _f32c(DZ_hat_j, (p, K)), _f32c(DZ_exp_logsum_hat, (p, K)),
         _f32c(log_U_hat), _f32c(log_V_hat, (p, K)), _f32c(D_hat, (n, p)), _f32c(X, (n, p))]
    rc = _lib().zq_zigap(*[_p(x) for x in a], n, p, K, 1 if quirk else 0)
    assert rc == 0


def zq_sparse_gap(SZ_hat_i, Z_hat_j, Z_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, X):
    """sparse_gap.py:81-97"""
    n, K = log_U_hat.shape
    p = log_V_hat.shape[0]
    a = [_f32c(SZ_hat_i, (n, K)), _f32c(Z_hat_j, (p, K)), _f32c(Z_exp_logsum_hat, (p, K)),
         _f32c(log_U_hat), _f32c(log_V_hat, (p, K)), _f32c(S_tilde, (p, K)), _f32c(S_hat, (p, K)),
         _f32c(X, (n, p))]
    rc = _lib().zq_sparse_gap(*[_p(x) for x in a], n, p, K)
    assert rc == 0


def zq_sparse_zigap(DSZ_hat, DZ_hat, DZ_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat,
                    D_hat, X):
    """sparse_zigap.py:100-116"""
    n, K = log_U_hat.shape
    p = log_V_hat.shape[0]
    a = [_f32c(DSZ_hat, (n, K)), _f32c(DZ_hat, (p, K)), _f32c(DZ_exp_logsum_hat, (p, K)),
         _f32c(log_U_hat), _f32c(log_V_hat, (p, K)), _f32c(S_tilde, (p, K)), _f32c(S_hat, (p, K)),
         _f32c(D_hat, (n, p)), _f32c(X, (n, p))]
    rc = _lib().zq_sparse_zigap(*[_p(x) for x in a], n, p, K)
    assert rc == 0


# ----------------------------------------------------------------------------------------
# utils.py
# ----------------------------------------------------------------------------------------
def zq_exact(log_U_hat, log_V_hat, X, S_tilde=None, S_hat=None, D_hat=None, quirk=False):
    """The four loop nests evaluated in float64 from the same float32 inputs ("exact": what the statements of
    gap.py:72-80 / zigap.py:84-95 / sparse_gap.py:86-97 / sparse_zigap.py:105-116 define before any float32 rounding).
    Returns float64 (Z_i, Z_j, Z_log).  Dense n x m x K intermediate: golden sizes only.  Used to measure how far the
    reference's own float32 evaluation and the HIP path each are from the exact value (tests, tools/parity_report.py)."""
    lu = np.asarray(log_U_hat, dtype=np.float64); lv = np.asarray(log_V_hat, dtype=np.float64)
    Xd = np.asarray(X, dtype=np.float64)
    K = lu.shape[1]
    with np.errstate(all='ignore'):
        ls = lu[:, None, :] + lv[None, :, :]
        e = np.exp(ls)
        if S_tilde is not None:
            e = e * np.asarray(S_tilde, dtype=np.float64)[None, :, :]
        den = e.sum(axis=2)
        den = np.where(den > 0, den, 1.0)
        r = Xd[:, :, None] * e / den[:, :, None]
        r = np.where(Xd[:, :, None] == 0, 0.0, r)                     # x = 0 contributes an exact 0 (0 * inf never arises in float32 either)
        d = np.asarray(D_hat, dtype=np.float64) if D_hat is not None else None
        wi = r if d is None else d[:, :, None] * r
        Zi = (wi * np.asarray(S_hat, dtype=np.float64)[None, :, :]).sum(axis=1) if S_hat is not None else wi.sum(axis=1)
        if d is not None and quirk:
            Zj = (d[:, :K][:, None, :] * r).sum(axis=0)               # zigap.py:94: D_hat[i, k]
        else:
            Zj = wi.sum(axis=0)
        Zlog = (wi * np.where(r == 0, 0.0, ls)).sum(axis=0)
    return Zi, Zj, Zlog


def logit(x):
    """utils.py:9-11"""
    x = np.clip(x, 1e-15, 1. - 1e-15)
    return np.log(x / (1. - x))


def sigmoid(x):
    """utils.py:14-15"""
    with np.errstate(over='ignore'):
        return 1. / (1. + np.exp(-x))


def digamma(x):
    """utils.py:31-32"""
    return scipy.special.digamma(x)


def digamma_prime(x):
    """utils.py:35-36"""
    return scipy.special.polygamma(1, x)


def inverse_digamma(y):
    """utils.py:39-51 (Minka's initialisation + 5 Newton steps)."""
    y = np.asarray(y, dtype=np.float64)
    with np.errstate(all='ignore'):
        x = np.where(y >= -2.22, np.exp(y) + .5, -1. / (y - digamma(1)))
        for _ in range(5):
            x = x - ((digamma(x) - y) / digamma_prime(x))
    return x


def clamp(a):
    """`np.maximum(1e-15, np.nan_to_num(a))` -- e.g. gap.py:99-100."""
    return np.maximum(EPS, np.nan_to_num(a))


# ----------------------------------------------------------------------------------------
# node expectations
# ----------------------------------------------------------------------------------------
def gamma_mean(a, b):
    """nodes/probabilistic/gamma.py:37-46 for the 'd,d' relations used by the models."""
    return np.asarray(a, dtype=np.float64) / np.asarray(b, dtype=np.float64)


def gamma_meanlog(a, b):
    """nodes/probabilistic/gamma.py:52-61: parameters are cast to f32 FIRST."""
    a = np.asarray(a).astype(np.float32)
    b = np.asarray(b).astype(np.float32)
    with np.errstate(all='ignore'):
        return digamma(a) - np.log(b)          # float32 result


def bernoulli_mean(p):
    """nodes/probabilistic/bernoulli.py:41-48"""
    return np.asarray(p).astype(np.float32)


# ----------------------------------------------------------------------------------------
# models
# ----------------------------------------------------------------------------------------
class _OracleModel:
    """Shared sequencing: models/base.py:43-56."""

    zi = False       # has the dropout node D (zigap.py, sparse_zigap.py)
    sparse = False   # has the sparsity node S (sparse_gap.py, sparse_zigap.py)
    skip_zeros = False   # True (pCMF / sparse pCMF): the loop nest through zq_gap_nz / zq_sparse_gap_nz -- bit-identical while no
                         # exponential overflows (tests/test_oracle.py), ~25 x faster at 4 % non-zeros
    exact = False    # True: the loop nest in float64 (zq_exact) instead of the reference's float32 -- the yardstick both the
                     # reference and the HIP path are measured against (tools/parity_report.py); everything else unchanged

    def __init__(self, X, k, init_a1, init_b1, tau=0.5, reference_quirks=True):
        self.X = np.asarray(X)
        self.Xf = np.ascontiguousarray(self.X.astype(np.float32))   # gap.py:94
        self.n, self.m = self.X.shape
        self.p = self.m
        self.k = int(k)
        self.tau = tau
        self.reference_quirks = reference_quirks
        K = self.k
        # build_u_node / build_v_node: the prior values are overwritten by the first
        # M-step (base.py:52) before anything reads them, except alpha2/beta2 == 1.
        self.alpha1 = np.ones(K)
        self.alpha2 = np.ones(K)
        self.beta1 = np.ones(K)
        self.beta2 = np.ones(K)
        # initialize_variational_parameters (gap.py:46-65 and twins)
        self.a1 = clamp(np.array(init_a1, dtype=np.float64).reshape(self.n, K))
        self.a2 = np.ones((self.n, K))
        self.b1 = clamp(np.array(init_b1, dtype=np.float64).reshape(self.m, K))
        self.b2 = np.ones((self.m, K))
        if self.zi:
            self.pi_d = np.zeros(self.p)
            self.p_d = (self.X > 0).astype(np.float64)               # zigap.py:77
        if self.sparse:
            self.pi_s = np.zeros(self.m)
            self.p_s = np.ones((self.m, K))                          # sparse_gap.py:79
        self.update_expectations()                                   # base.py:49
        self.update_prior_hyper_parameters()                         # base.py:52

    # ---- expectations (gap.py:131-135, zigap.py:160-165, sparse_gap.py:167-172) ----
    def update_expectations(self):
        self.U_hat = gamma_mean(self.a1, self.a2)
        self.V_hat = gamma_mean(self.b1, self.b2)          # Vprime_hat in the sparse models
        self.log_U_hat = gamma_meanlog(self.a1, self.a2)
        self.log_V_hat = gamma_meanlog(self.b1, self.b2)   # log_Vprime_hat in the sparse models
        if self.zi:
            self.D_hat = bernoulli_mean(self.p_d)
        if self.sparse:
            self.S_hat = bernoulli_mean(self.p_s)

    def step(self):
        """base.py:54-56"""
        self.update_variational_parameters()
        self.update_prior_hyper_parameters()

    # ---- M-step (gap.py:117-129, zigap.py:143-158, sparse_gap.py:150-165,
    #      sparse_zigap.py:178-196) ----
    def update_prior_hyper_parameters(self):
        with np.errstate(all='ignore'):
            self.alpha1 = clamp(inverse_digamma(np.log(self.alpha2) + np.mean(self.log_U_hat, axis=0)))
            self.alpha2 = clamp(self.alpha1 / np.mean(self.U_hat, axis=0))
            self.beta1 = clamp(inverse_digamma(np.log(self.beta2) + np.mean(self.log_V_hat, axis=0)))
            self.beta2 = clamp(self.beta1 / np.mean(self.V_hat, axis=0))
        if self.zi:
            self.pi_d = np.mean(self.p_d, axis=0)
        if self.sparse:
            self.pi_s = np.mean(self.p_s, axis=1)

    # ---- E-step ----
    def update_variational_parameters(self):
        n, m, K = self.n, self.m, self.k
        Zi = np.empty((n, K), dtype=np.float32)
        Zj = np.empty((m, K), dtype=np.float32)
        Zlog = np.empty((m, K), dtype=np.float32)
        if self.sparse:
            S_tilde = (self.p_s > self.tau).astype(np.float32)       # sparse_gap.py:113
        if self.exact:
            Zi, Zj, Zlog = zq_exact(self.log_U_hat, self.log_V_hat, self.Xf, S_tilde if self.sparse else None,
                                    self.S_hat if self.sparse else None, self.D_hat if self.zi else None,
                                    quirk=(self.zi and not self.sparse and self.reference_quirks))
        elif not self.zi and not self.sparse:
            (zq_gap_nz if self.skip_zeros else zq_gap)(Zi, Zj, self.log_U_hat, self.log_V_hat, self.Xf)   # gap.py:89-94
        elif self.zi and not self.sparse:
            zq_zigap(Zi, Zj, Zlog, self.log_U_hat, self.log_V_hat, self.D_hat, self.Xf,
                     quirk=self.reference_quirks)                                 # zigap.py:105-112
        elif self.sparse and not self.zi:
            (zq_sparse_gap_nz if self.skip_zeros else zq_sparse_gap)(Zi, Zj, Zlog, self.log_U_hat, self.log_V_hat, S_tilde,
                                                                     self.S_hat, self.Xf)   # sparse_gap.py:107-115
        else:
            zq_sparse_zigap(Zi, Zj, Zlog, self.log_U_hat, self.log_V_hat, S_tilde,
                            self.S_hat, self.D_hat, self.Xf)                      # sparse_zigap.py:126-135
        self.last_Z = (Zi, Zj, Zlog)
        if self.zi:
            self.last_D_hat = self.D_hat        # the D_hat this sweep's sums used (tests: conditioning)

        with np.errstate(all='ignore'):
            # -- U_q (gap.py:96-102, zigap.py:114-120, sparse_gap.py:117-124, sparse_zigap.py:137-144)
            V_eff = self.S_hat * self.V_hat if self.sparse else self.V_hat
            self.a1 = clamp(self.alpha1[np.newaxis, :] + Zi)
            if self.zi:
                self.a2 = clamp(self.alpha2 + np.dot(self.D_hat, V_eff))
            else:
                self.a2 = clamp(np.broadcast_to(self.alpha2 + V_eff.sum(axis=0), (n, K)).copy())
            self.U_hat = gamma_mean(self.a1, self.a2)
            self.log_U_hat = gamma_meanlog(self.a1, self.a2)

            # -- V_q / Vprime_q (gap.py:104-110, zigap.py:122-128, sparse_gap.py:126-132,
            #    sparse_zigap.py:146-152).  sparse_gap.py:127 reads a bare `S_hat`
            #    (NameError in the reference); the evident intent self.S_hat is used.
            if self.zi:
                c = np.dot(self.D_hat.T, self.U_hat)                 # (m, K), new U_hat
            else:
                c = np.broadcast_to(self.U_hat.sum(axis=0), (m, K))
            if self.sparse:
                self.b1 = clamp(self.beta1[np.newaxis, :] + self.S_hat * Zj)
                self.b2 = clamp(self.beta2 + self.S_hat * c)
            else:
                self.b1 = clamp(self.beta1[np.newaxis, :] + Zj)
                self.b2 = clamp(self.beta2 + c)
            self.V_hat = gamma_mean(self.b1, self.b2)
            self.log_V_hat = gamma_meanlog(self.b1, self.b2)

            # -- S_q (sparse_gap.py:134-141, sparse_zigap.py:154-161)
            if self.sparse:
                tmp = -Zlog
                tmp = tmp + np.nan_to_num(c * self.V_hat)
                p_s = sigmoid(logit(self.pi_s)[:, np.newaxis] - tmp)
                p_s = np.nan_to_num(p_s)
                p_s[self.pi_s <= 0] = 1e-10
                p_s[self.pi_s >= 1] = 1. - 1e-10
                self.p_s = p_s
                S_hat_new = bernoulli_mean(self.p_s)

            # -- D_q (zigap.py:130-136, sparse_zigap.py:163-169): sparse_zigap uses the
            #    V_hat = S_hat * Vprime_hat computed BEFORE the updates (line 138).
            if self.zi:
                V_for_d = V_eff if self.sparse else self.V_hat
                p_d = sigmoid(logit(self.pi_d)[np.newaxis, :] - np.dot(self.U_hat, V_for_d.T))
                p_d[:, self.pi_d <= 0] = 1e-10
                p_d[:, self.pi_d >= 1] = 1. - 1e-10
                p_d[self.X != 0] = 1. - 1e-10
                self.p_d = p_d
                self.D_hat = bernoulli_mean(self.p_d)
            if self.sparse:
                self.S_hat = S_hat_new

    # ---- metrics (base.py:58-87 with loglikelihood_X, sparse_zigap.py:44-51) ----
    # The reference defines loglikelihood_X on SparseZIGaP only, so the deviances raise AttributeError
    # on the other three classes; here the absent nodes read as constants (pi_d = 1, D_hat = 1,
    # S_hat = 1), which is what the SparseZIGaP formulas reduce to.  float64 throughout: the reference
    # stores the per-entry terms in `np.empty_like(X)` (sparse_zigap.py:45) and therefore truncates
    # them when X is an integer array; with a float X it computes exactly this.
    def _metric_inputs(self):
        pi_d = self.pi_d if self.zi else np.ones(self.p)
        D = np.round(self.D_hat) if self.zi else np.ones((self.n, self.p))           # base.py:60
        V = self.V_hat * self.S_hat if self.sparse else self.V_hat                   # base.py:65
        return pi_d, D, np.dot(self.U_hat, V.T)                                      # base.py:66

    def loglikelihood_X(self, Lambda, pi_d):
        """sparse_zigap.py:44-51"""
        X = self.X.astype(np.float64)
        ret = np.empty_like(X)
        pi = np.repeat(pi_d[np.newaxis, ...], X.shape[0], axis=0)
        z = X == 0
        with np.errstate(all='ignore'):
            ret[z] = np.log(pi[z] * np.exp(-Lambda[z]) + (1 - pi[z]))
            ret[~z] = np.log(pi[~z]) - Lambda[~z] + X[~z] * np.log(Lambda[~z])
        return ret.sum()

    def reconstruction_deviance(self):
        """base.py:58-69"""
        pi_d, D, Lambda = self._metric_inputs()
        mask = D == 0
        assert not mask.all()
        X = self.X.astype(np.float64)
        ll_X_given_X = self.loglikelihood_X(X, pi_d)
        Lambda[mask] = 0
        ll_X_given_UV = self.loglikelihood_X(Lambda, pi_d)
        return -2. * (ll_X_given_UV - ll_X_given_X)

    def explained_deviance(self):
        """base.py:71-82, called after reconstruction_deviance (experiments/clustering.py:26-27):
        the U, V, D node buffers are the ones that call left behind."""
        pi_d, D, Lambda = self._metric_inputs()
        mask = D == 0
        assert not mask.all()
        X = self.X.astype(np.float64)
        ll_X_given_X = self.loglikelihood_X(X, pi_d)
        mean = np.repeat(X.mean(axis=0)[np.newaxis, ...], X.shape[0], axis=0)
        ll_X_given_X_mean = self.loglikelihood_X(mean, pi_d)
        Lambda[mask] = 0
        ll_X_given_UV = self.loglikelihood_X(Lambda, pi_d)
        return (ll_X_given_UV - ll_X_given_X_mean) / (ll_X_given_X - ll_X_given_X_mean)

    def frobenius_norm(self):
        """base.py:84-87 on the UV buffer the deviance calls leave behind (Lambda with the mask applied)."""
        _, D, Lambda = self._metric_inputs()
        Lambda[D == 0] = 0
        return np.sqrt(((Lambda.flatten() - self.X.astype(np.float64).flatten()) ** 2.).sum())

    def state(self):
        keys = ['alpha1', 'alpha2', 'beta1', 'beta2', 'a1', 'a2', 'b1', 'b2',
                'U_hat', 'V_hat', 'log_U_hat', 'log_V_hat']
        if self.zi:
            keys += ['pi_d', 'p_d', 'D_hat']
        if self.sparse:
            keys += ['pi_s', 'p_s', 'S_hat']
        return {k: np.array(getattr(self, k)) for k in keys}

    def load_state(self, st):
        for k, v in st.items():
            if hasattr(self, k):
                setattr(self, k, np.array(v))


class OracleGaP(_OracleModel):
    """models/gap.py:14-135"""


class OracleZIGaP(_OracleModel):
    """models/zigap.py:15-165"""
    zi = True


class OracleSparseGaP(_OracleModel):
    """models/sparse_gap.py:15-172 (with the sparse_gap.py:127 NameError repaired)."""
    sparse = True


class OracleSparseZIGaP(_OracleModel):
    """models/sparse_zigap.py:15-204"""
    zi = True
    sparse = True


MODELS = {'GaP': OracleGaP, 'ZIGaP': OracleZIGaP, 'SparseGaP': OracleSparseGaP,
          'SparseZIGaP': OracleSparseZIGaP}

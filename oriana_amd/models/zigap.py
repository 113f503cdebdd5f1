# -*- coding: utf-8 -*-
"""Zero-inflated and sparse variants (reference oriana/models/zigap.py:15-165,
sparse_gap.py:15-172, sparse_zigap.py:15-204) on the HIP path.

The responsibility sums run on the resident non-zero tiles (engine.zq); the dropout posterior is
exactly 1 (in float32) at every non-zero count -- zigap.py:135 sets p_d[X != 0] = 1 - 1e-10 and
Bernoulli.mean casts to float32 (bernoulli.py:45) -- so the loop nests never need D_hat[i, j] at
the non-zeros.  The three dense contractions of the ZI models (D_hat V_hat, D_hat^T U_hat,
U_hat V_hat^T: zigap.py:116, 124, 132) run on the matrix cores.  Inside a sweep (csrc/dense_f32.hip, K <= 128):
float32 products whose long sums end in float64; U_hat V_hat^T is fused with the sigmoid / override / column-sum
epilogue so that Lambda is never materialised, and the same kernel forms D_hat V_hat for the NEXT sweep from the
tile of D_hat it is about to store -- a sweep reads D_hat once (D_hat^T U_hat) and writes it once.  The float64
kernels (csrc/dense_mfma.hip) evaluate p_d itself on access, the metrics, K > 128, and D_hat V_hat whenever the
product kept from the previous sweep does not apply (first sweep, state written from outside);
ORIANA_ZI_EXACT=1 routes everything through them.
"""
import os

import numpy as np
import torch

from .. import engine
from .. import dist as odist
from .._lib import call, ptr, stream_ptr
from ..parameters import Parameter, LazyParameter
from .base import FactorModel

__all__ = ['ZIGaP', 'SparseGaP', 'SparseZIGaP']


class _ZIMixin:
    """Dropout node D (zigap.py:31-43, 76-77, 130-136, 157-158)."""

    def _init_zi(self):
        n, m, dev = self.n, self.m, self.device
        # The dense ZI kernels move 16-byte pieces of D_hat rows, masks and logits (csrc/dense_zi.hip, dense_f32.hip): the
        # gene axis of D_hat, of the non-zero mask and of every per-gene operand they take is padded to a multiple of 4
        # with INERT genes -- no counts, V_hat row 0, pi_d 0 (so p_d = 1e-10 by the column override of zigap.py:133): they add
        # nothing to D_hat V_hat, their rows of D_hat^T U_hat and of the column sums are dropped.  m % 4 == 0: nothing changes.
        mp = self._mp = (m + 3) // 4 * 4
        self._padbuf = {}
        self.pi_d = Parameter(torch.zeros(m, dtype=torch.float64, device=dev))
        # p_d = (X > 0) as float (zigap.py:77): exactly 1.0 at the non-zero counts, so D_hat holds it
        # exactly and the float64 matrix is only evaluated on access (LazyParameter)
        self._Dp = torch.zeros(n, mp, dtype=torch.float32, device=dev)
        self._D_hat = self._Dp[:, :m]                   # (n, m) view: what the model's own arithmetic and the API see
        ct = self.counts
        call('oriana_dropout_fix_nz_ld', ct.sparse_struct, None, ptr(self._Dp), 1.0, mp, stream_ptr())
        if ct.dense is not None:           # hybrid layout: the non-zero counts of the dense genes
            call('oriana_dense_fix_nz', ct.dense.c_struct, ptr(self._Dp), mp, ptr(ct.row_perm), ptr(ct.col_perm), 1.0, stream_ptr())
        self.p_d = LazyParameter((n, m), dev, lambda: self._D_hat.double())
        self._pd_sum_p = torch.zeros(mp, dtype=torch.float64, device=dev)
        self._pd_sum = self._pd_sum_p[:m]
        # bit mask of X != 0 (constant): lets the D update apply p_d[X != 0] = 1 - 1e-10 in its own pass
        self._nzmask = torch.zeros(((n + 31) // 32) * max(mp, 1), dtype=torch.int32, device=dev)
        call('oriana_nzmask_f32', ptr(self._nzmask), ptr(self._Dp), n, mp, stream_ptr())
        # [r6] the same mask as per-lane flags for the K = 33 .. 100 D-update kernel (csrc/dense_zi.hip)
        from .. import _lib
        self._nztiles = torch.zeros(max(int(_lib.load().oriana_nzmask_tiles_words(n, mp)), 4), dtype=torch.int32, device=dev)
        call('oriana_nzmask_tiles', ptr(self._nztiles), ptr(self._nzmask), n, mp, stream_ptr())
        self._pd_sum_fresh = False
        # non-zero counts per gene (local rows): the float32 sweep kernel counts p_d = float32(1 - 1e-10) = 1 at the non-zeros;
        # the M-step takes the 1e-10 per entry back, as the reference's float64 mean has it (zigap.py:135, 158) -- a gene
        # expressed in every cell then gets pi_d = 1 - 1e-10 (finite logit), not 1
        self._nnz_gene_p = torch.zeros(mp, dtype=torch.float64, device=dev)
        self._nnz_gene = self._nnz_gene_p[:m]
        call('oriana_colsum_wide_f32', ptr(self._nnz_gene_p), ptr(self._Dp), n, mp, stream_ptr())
        # float32 matrix-core path of the sweep (dense_f32.hip) and the D_hat V product it leaves for the next sweep,
        # valid while (D_hat, V_hat, S_hat) are the tensors it was formed from: _ver counts their writes
        self._fast_dense = self.k <= 128 and os.environ.get('ORIANA_ZI_EXACT', '0') != '1'
        self._DV_next = None
        self.n_kept_products = 0          # sweeps whose D_hat V came from the previous sweep's D update
        self._lg_scratch = torch.zeros(int(_lib.load().oriana_dropout_sweep_scratch_floats(mp, self.k)), dtype=torch.float32, device=dev)
        self._dt_scratch = torch.zeros(int(_lib.load().oriana_dense_t_scratch_floats(n, self.k)), dtype=torch.float32, device=dev)
        # how the float32 products are evaluated (include/oriana_hip.h): 1 = three-way bf16 splits on the bf16 matrix
        # cores (K <= 100), 0 = the float32 matrix instruction
        self._matrix_arith = {'f32': 0, 'bf16x3': 1}[os.environ.get('ORIANA_ZI_MATRIX', 'bf16x3')]

    def _padG(self, T, key):
        """A per-gene operand ((m, K) or (m,) float64) with the inert genes appended as zeros; T itself when m % 4 == 0."""
        if T is None:
            return None
        if self._mp == self.m:
            return T.contiguous()
        buf = self._padbuf.get(key)
        shape = (self._mp,) + tuple(T.shape[1:])
        if buf is None or buf.shape != shape or buf.dtype != T.dtype:
            buf = self._padbuf[key] = torch.zeros(shape, dtype=T.dtype, device=self.device)
        buf[:self.m].copy_(T)
        return buf

    @property
    def D_hat(self):
        return self._D_hat.cpu().numpy()

    def _refresh_D_hat(self):
        """Bernoulli.mean: float32 cast of p_d (bernoulli.py:45).  While p_d is not materialised it is,
        by construction, what D_hat was cast from."""
        if self.p_d.materialised:
            self._D_hat.copy_(self.p_d.tensor)
            self._pd_sum_fresh = False
            self._DV_next = None

    def _mstep_pi_d(self):
        """pi_d = mean(p_d, axis=0) (zigap.py:158), summed over the row shards.  The column sums
        come for free from the D update of the same sweep; otherwise (initialisation, p_d written
        from outside, two M-steps in a row) they are taken from whatever currently defines p_d."""
        if not self._pd_sum_fresh:
            self._pd_sum.zero_()
            st = stream_ptr()
            if self.p_d.materialised:
                call('oriana_colsum_wide_f64', ptr(self._pd_sum), ptr(self.p_d.tensor), self.n, self.m, st)
            elif self._pd_snap is None:                 # p_d == D_hat exactly (zigap.py:77)
                call('oriana_colsum_wide_f32', ptr(self._pd_sum_p), ptr(self._Dp), self.n, self._mp, st)
            else:                                       # re-evaluate the sums only, nothing is stored
                U, V, pi_d = self._pd_snap
                call('oriana_dropout_update_fused', None, None, ptr(U), ptr(self._padG(V, 'Vs')), ptr(self._padG(pi_d, 'pis')),
                     ptr(self._nzmask), ptr(self._pd_sum_p), self.n, self._mp, self.k, st)
        self._pd_sum_fresh = False
        odist.all_reduce_sum(self._pd_sum, self.pg)
        torch.div(self._pd_sum, float(self.n_total), out=self.pi_d.tensor)

    _pd_snap = None

    def _D_times(self, V):
        """np.dot(D_hat, V): float32 D_hat promoted to float64 (zigap.py:116).  (n, K).  Inside a run of sweeps this is
        the product the previous sweep's D update left behind (_update_D); otherwise f64 MFMA."""
        kept = self._DV_next
        self._DV_next = None
        if kept is not None and kept[1] == self._ver:
            self.n_kept_products += 1
            return kept[0]
        out = torch.zeros(self.n, self.k, dtype=torch.float64, device=self.device)
        with engine._span(self._ws, 'DV'):
            call('oriana_dense_times_factor', ptr(out), ptr(self._Dp), ptr(self._padG(V, 'Vt')), self.n, self._mp, self.k, 0,
                 stream_ptr())
        return out

    def _Dt_times(self, U):
        """np.dot(D_hat.T, U) (zigap.py:124) over the LOCAL rows: the shards' partials are summed by the sweep's
        packed exchange.  (m, K)."""
        out = torch.zeros(self._mp, self.k, dtype=torch.float64, device=self.device)
        with engine._span(self._ws, 'DtU'):
            if self._fast_dense:
                call('oriana_dense_t_times_factor_f32', ptr(out), ptr(self._Dp), ptr(U.contiguous()), ptr(self._dt_scratch),
                     self._matrix_arith, self.n, self._mp, self.k, stream_ptr())
            else:
                call('oriana_dense_times_factor', ptr(out), ptr(self._Dp), ptr(U.contiguous()), self.n, self._mp,
                     self.k, 1, stream_ptr())
        return out[:self.m]                              # (the inert genes' rows are dropped)

    def _update_D(self, V_for_d, V_next=None):
        """zigap.py:130-136: p_d = sigmoid(logit(pi_d) - U_hat V^T), overrides, D_hat; one fused
        kernel that stores D_hat and leaves the column sums of p_d for the pi_d M-step.  The float64
        p_d itself is not stored: it is re-evaluated on access from a snapshot of the three inputs.
        V_next: the factor the next sweep's cell update multiplies D_hat with, as it stands now."""
        self._pd_sum_p.zero_()
        V = V_for_d.contiguous()
        Vp, pip = self._padG(V, 'Vd'), self._padG(self.pi_d.tensor, 'pid')
        with engine._span(self._ws, 'D_update'):
            if self._fast_dense:
                DV = None
                if V_next is not None:
                    DV = torch.zeros(self.n, self.k, dtype=torch.float64, device=self.device)
                    V_next = Vp if V_next is V_for_d else self._padG(V_next, 'Vn')
                call('oriana_dropout_sweep_fused_tiles', ptr(self._Dp), ptr(self._U_hat), ptr(Vp), ptr(pip),
                     ptr(self._nzmask), ptr(self._nztiles), ptr(self._pd_sum_p), ptr(V_next), ptr(DV), ptr(self._lg_scratch),
                     self._matrix_arith, self.n, self._mp, self.k, stream_ptr())
                self._pd_sum.sub_(self._nnz_gene, alpha=1e-10)      # the non-zeros are 1 - 1e-10 each, not 1
            else:
                DV = None
                call('oriana_dropout_update_fused', None, ptr(self._Dp), ptr(self._U_hat), ptr(Vp), ptr(pip),
                     ptr(self._nzmask), ptr(self._pd_sum_p), self.n, self._mp, self.k, stream_ptr())
        self._touch()
        self._DV_next = (DV, self._ver) if DV is not None else None
        self._pd_sum_fresh = True
        self._pd_snap = snap = (self._U_hat.clone(), V.clone(), self.pi_d.tensor.clone())
        self.p_d.defer(lambda: self._evaluate_p_d(*snap))

    def _evaluate_p_d(self, U, V, pi_d):
        p_d = torch.empty(self.n, self._mp, dtype=torch.float64, device=self.device)
        call('oriana_dropout_update_fused', ptr(p_d), None, ptr(U), ptr(self._padG(V, 'Vs')), ptr(self._padG(pi_d, 'pis')),
             ptr(self._nzmask), None, self.n, self._mp, self.k, stream_ptr())
        return p_d if self._mp == self.m else p_d[:, :self.m].contiguous()


class _SparseMixin:
    """Sparsity node S on V (sparse_gap.py:26-34, 79, 113, 134-141, 164-165)."""

    def _init_sparse(self):
        m, K, dev = self.m, self.k, self.device
        self.pi_s = Parameter(torch.zeros(m, dtype=torch.float64, device=dev))
        self.p_s = Parameter(torch.ones(m, K, dtype=torch.float64, device=dev))       # sparse_gap.py:79
        self._S_hat = torch.ones(m, K, dtype=torch.float32, device=dev)
        self._S_tilde = torch.ones(m, K, dtype=torch.float32, device=dev)
        self._Zlog = self._xch.f32['Zlog']          # a view into the sweep's packed exchange buffer
        self._Veff = torch.zeros(m, K, dtype=torch.float64, device=dev)
        self._sumVeff = torch.zeros(K, dtype=torch.float64, device=dev)

    @property
    def S_hat(self):
        return self._S_hat.cpu().numpy()

    # the Gamma node of the sparse models is Vprime (sparse_gap.py:31): same buffers, reference names
    @property
    def Vprime_hat(self):
        return self.V_hat

    @property
    def log_Vprime_hat(self):
        return self.log_V_hat

    def _effective_V(self):
        self._compute_Veff()
        return self._Veff

    def _compute_Veff(self):
        """V_hat = S_hat * Vprime_hat (sparse_gap.py:118)."""
        call('oriana_mul_f64_f32', ptr(self._Veff), ptr(self._V_hat), ptr(self._S_hat), self.m * self.k, stream_ptr())

    def _refresh_S_hat(self):
        self._S_hat.copy_(self.p_s.tensor)
        self._touch()

    def _mstep_pi_s(self):
        call('oriana_rowmean_f64', ptr(self.pi_s.tensor), ptr(self.p_s.tensor), self.m, self.k, stream_ptr())

    def _update_S(self, c_vec=None, c_mat=None):
        """sparse_gap.py:134-141 (uses the NEW Vprime_hat and the sums of the NEW U_hat)."""
        call('oriana_sparsity_update', ptr(self.p_s.tensor), ptr(self._S_hat), ptr(self.pi_s.tensor), ptr(self._Zlog),
             ptr(c_vec), ptr(c_mat), ptr(self._V_hat), self.m, self.k, stream_ptr())
        self._touch()


class ZIGaP(_ZIMixin, FactorModel):
    """ZI-pCMF (reference zigap.py:15-165)."""
    zi = True

    @staticmethod
    def compute_Z_q_expectations(DZ_hat_i, DZ_hat_j, DZ_exp_logsum_hat, log_U_hat, log_V_hat, D_hat, X,
                                 reference_quirks=True):
        """Drop-in for zigap.py:79-95 on dense float32 device tensors (outputs first, zero-filled
        here, returns None).  reference_quirks keeps the D_hat[i, k] index of zigap.py:94."""
        engine.zq_dense(DZ_hat_i, DZ_hat_j, DZ_exp_logsum_hat, log_U_hat, log_V_hat, X, D_hat=D_hat,
                        quirk=reference_quirks)

    def _init_extra(self):
        self._init_zi()

    def update_expectations(self):
        FactorModel.update_expectations(self)
        self._refresh_D_hat()

    def update_prior_hyper_parameters(self):
        FactorModel.update_prior_hyper_parameters(self)
        self._mstep_pi_d()

    def update_variational_parameters(self):
        """zigap.py:97-141."""
        dq = None
        if self.reference_quirks:
            # zigap.py:94 weights the per-gene sums with D_hat[i, k] (first K gene columns)
            dq = torch.empty(self.n, self.k, dtype=torch.float32, device=self.device)
            call('oriana_take_cols_f32', ptr(dq), ptr(self._Dp), self.n, self._mp, self.k, stream_ptr())
        zq_args = (self._ws, self._Zi, self._Zj, None, self._log_U_hat, self._log_V_hat)
        engine.zq(*zq_args, dq=dq, phase='rows')
        # U_q: a2 = alpha2 + D_hat V_hat (OLD V_hat)                                  zigap.py:115-120
        self._gamma_side('u', self._Zi, rate_mat=self._D_times(self._V_hat))
        DtU = self._Dt_times(self._U_hat)           # local rows, NEW U_hat
        self._exchange_start(DtU=DtU)               # D_hat^T U_hat | column sums of U_hat (float64): reduced under the column pass
        engine.zq(*zq_args, dq=dq, phase='cols')
        DtU = self._exchange()['DtU']               # Z_j (float32) + wait for the float64 partials
        # V_q: b2 = beta2 + D_hat^T U_hat (NEW U_hat)                                 zigap.py:123-128
        self._gamma_side('v', self._Zj, rate_mat=DtU)
        # D_q (NEW U_hat, NEW V_hat), and D_hat V_hat for the next sweep's U_q         zigap.py:130-136, 116
        self._update_D(self._V_hat, V_next=self._V_hat)

    def _load_extra(self, st):
        self._DV_next = None
        if 'p_d' in st:
            self._refresh_D_hat()


class SparseGaP(_SparseMixin, FactorModel):
    """Sparse pCMF (reference sparse_gap.py:15-172; the NameError of sparse_gap.py:127 -- a bare
    `S_hat` -- is read as the evident self.S_hat, SURVEY.md 8(a) policy)."""
    sparse = True

    @staticmethod
    def compute_Z_q_expectations(SZ_hat_i, Z_hat_j, Z_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, X):
        """Drop-in for sparse_gap.py:81-97 on dense float32 device tensors."""
        engine.zq_dense(SZ_hat_i, Z_hat_j, Z_exp_logsum_hat, log_U_hat, log_V_hat, X, S_tilde=S_tilde, S_hat=S_hat)

    def _init_extra(self):
        self._init_sparse()

    def update_expectations(self):
        FactorModel.update_expectations(self)
        self._refresh_S_hat()

    def update_prior_hyper_parameters(self):
        FactorModel.update_prior_hyper_parameters(self)
        self._mstep_pi_s()

    def _threshold(self):
        call('oriana_threshold_f32', ptr(self._S_tilde), ptr(self.p_s.tensor), float(self.tau), self.m * self.k, stream_ptr())

    def update_variational_parameters(self):
        """sparse_gap.py:99-148."""
        self._threshold()                                                              # sparse_gap.py:113
        zq_args = (self._ws, self._Zi, self._Zj, self._Zlog, self._log_U_hat, self._log_V_hat)
        engine.zq(*zq_args, S_tilde=self._S_tilde, S_hat=self._S_hat, phase='rows')
        # U_q: a2 = alpha2 + sum_j S_hat * Vprime_hat (OLD)                            sparse_gap.py:118-124
        self._sumVeff.zero_()
        call('oriana_colsum_f64', ptr(self._sumVeff), ptr(self._V_hat), ptr(self._S_hat), self.m, self.k, stream_ptr())
        self._gamma_side('u', self._Zi, rate_vec=self._sumVeff)
        self._exchange_start()                      # column sums of U_hat (float64): reduced under the column pass
        engine.zq(*zq_args, S_tilde=self._S_tilde, S_hat=self._S_hat, phase='cols')
        self._exchange()                            # Z_j | Z_log (float32) + wait for the sums
        # Vprime_q: b1 = beta1 + S_hat * Z_j ; b2 = beta2 + S_hat * sum_i U_hat (NEW)  sparse_gap.py:127-132
        self._gamma_side('v', self._Zj, zmul=self._S_hat, rate_vec=self._sumU[0], rmul=self._S_hat)
        # S_q                                                                          sparse_gap.py:134-141
        self._update_S(c_vec=self._sumU[0])

    def _load_extra(self, st):
        if 'p_s' in st:
            self._refresh_S_hat()


class SparseZIGaP(_ZIMixin, _SparseMixin, FactorModel):
    """Sparse ZI-pCMF (reference sparse_zigap.py:15-204)."""
    zi = True
    sparse = True

    @staticmethod
    def compute_Z_q_expectations(DSZ_hat, DZ_hat, DZ_exp_logsum_hat, log_U_hat, log_V_hat, S_tilde, S_hat, D_hat, X):
        """Drop-in for sparse_zigap.py:100-116 on dense float32 device tensors."""
        engine.zq_dense(DSZ_hat, DZ_hat, DZ_exp_logsum_hat, log_U_hat, log_V_hat, X, S_tilde=S_tilde, S_hat=S_hat,
                        D_hat=D_hat)

    def _init_extra(self):
        self._init_zi()
        self._init_sparse()

    def update_expectations(self):
        FactorModel.update_expectations(self)
        self._refresh_D_hat()
        self._refresh_S_hat()

    def update_prior_hyper_parameters(self):
        FactorModel.update_prior_hyper_parameters(self)
        self._mstep_pi_d()                                                             # sparse_zigap.py:193
        self._mstep_pi_s()                                                             # sparse_zigap.py:196

    _threshold = SparseGaP._threshold

    def update_variational_parameters(self):
        """sparse_zigap.py:118-176."""
        self._threshold()
        zq_args = (self._ws, self._Zi, self._Zj, self._Zlog, self._log_U_hat, self._log_V_hat)
        engine.zq(*zq_args, S_tilde=self._S_tilde, S_hat=self._S_hat, phase='rows')
        # V_hat = S_hat * Vprime_hat, computed BEFORE the updates and used again by the D update
        # (sparse_zigap.py:138, 165)
        self._compute_Veff()
        V_old = self._Veff.clone()
        self._gamma_side('u', self._Zi, rate_mat=self._D_times(V_old))                 # sparse_zigap.py:139-144
        DtU = self._Dt_times(self._U_hat)                                               # local rows, NEW U_hat, OLD D_hat
        self._exchange_start(DtU=DtU)               # the float64 partials: reduced under the column pass
        engine.zq(*zq_args, S_tilde=self._S_tilde, S_hat=self._S_hat, phase='cols')
        DtU = self._exchange()['DtU']               # Z_j | Z_log (float32) + wait for D_hat^T U_hat | column sums of U_hat
        self._gamma_side('v', self._Zj, zmul=self._S_hat, rate_mat=DtU, rmul=self._S_hat)   # :147-152
        self._update_S(c_mat=DtU)                                                       # :154-161
        # :163-169; the next sweep multiplies D_hat with S_hat * Vprime_hat as they stand now (:138)
        self._compute_Veff()
        self._update_D(V_old, V_next=self._Veff)

    def _load_extra(self, st):
        self._DV_next = None
        if 'p_d' in st:
            self._refresh_D_hat()
        if 'p_s' in st:
            self._refresh_S_hat()

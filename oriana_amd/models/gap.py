# -*- coding: utf-8 -*-
"""pCMF = Gamma-Poisson factor model (reference oriana/models/gap.py:14-135)."""
import os

import torch

from .. import engine
from .base import FactorModel

__all__ = ['GaP']


class GaP(FactorModel):

    @staticmethod
    def compute_Z_q_expectations(Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X):
        """Drop-in for the reference's loop nest (gap.py:67-80): outputs first, caller allocates,
        callee zero-fills, returns None.  All arguments are 2-D C-contiguous float32 DEVICE tensors
        (TypeError otherwise, like numba's explicit signature); X is dense (n, m) and is repacked on
        every call through the stateless C entry oriana_zq_gap_f32 (the model itself keeps X packed
        across sweeps)."""
        engine.zq_gap_stateless(Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X)

    def update_variational_parameters(self):
        """gap.py:82-115 (E-step)."""
        # both Z sums use the PRE-update E[log U], E[log V] (one joint pass, gap.py:89-94); the cell side only
        # needs Z_i, so its update runs between the row pass and the column pass and every partial of the
        # sweep's single exchange exists when the column pass ends
        # Launch count matters on a small matrix (configs[1]: a sweep is ~10 launches of a few microseconds of work each):
        # the zero-fills ride on the factor preparation, Z += F * R on the Gamma updates, both M-steps share a launch.
        ws, ct = self._ws, self.counts
        if self._v_sums_in_acc:                # an E-step without the M-step before it: sum_j V_hat is still in scratch
            self._sumV.copy_(self._accV)
            self._v_sums_in_acc = False
        fold_cols = not self.sharded           # (under row sharding Z_j is completed per rank, then exchanged)
        # Row sharding, K == Kp: the per-gene sums are exchanged in the PACKED gene order, where the dense genes [0, gd) and
        # the sliced genes [gd, m) are contiguous segments -- the sliced segment (86 % of the 12 MB at C4) starts its
        # all-reduce as soon as the sliced column pass has finished and travels under the dense gene-side kernel; the
        # gene-side update then reads the reduced sums through the permutation (Z[o] = 1 * Zx[p], exact).
        packed = self.sharded and ws.Kp == self.k
        engine.zq_gap(ws, self._Zi, self._Zj, self._log_U_hat, self._log_V_hat, phase='rows', finalize_rows=False,
                      clear=(self._sumU, self._accV) + ((self._Zj_o,) if packed else ()), zj_packed=packed)
        # U_q: a1 = alpha1 + Z_i ; a2 = alpha2 + sum_j V_hat (OLD V_hat)                gap.py:97-102
        lazy = self._gamma_side_finalize('u', self._Zi, ws.FU, ws.R, ct.row_perm, self._sumV[0], self._sumU,
                                         nslab=ws.row_gene_splits, slab_row0=ws.row_slab_row0,
                                         a2_row=self._a2_row if self._lazy_ok else None)
        if lazy:
            self._u_on_access()
        else:
            self._lazy_ok = False            # (no vector kernel for this K / a launch-bound size: decided once per model)
        self._exchange_start()                  # sum_i U_hat | sum_i log U_hat (float64): reduced under the column pass
        engine.zq_gap(ws, self._Zi, self._Zj, self._log_U_hat, self._log_V_hat, phase='cols', finalize_cols=not fold_cols,
                      zj_packed=packed,
                      on_segment=(lambda lo, hi: self._xch.reduce_rows_async('Zj', lo, hi)) if packed else None)
        self._exchange()                        # Z_j (float32, 12 MB at C4; its segments are already on their way) + wait
        # V_q: b1 = beta1 + Z_j ; b2 = beta2 + sum_i U_hat (NEW U_hat)                   gap.py:105-110
        # (its column sums go to scratch: _sumV still holds sum_j V_hat of the sweep's start, which the M-step replaces)
        if fold_cols:
            self._gamma_side_finalize('v', self._Zj, ws.FV, ws.C, ct.col_perm, self._sumU[0], self._accV)
        elif packed:
            self._gamma_side_finalize('v', self._Zj_o, self._ones_v, self._Zj, ct.col_perm, self._sumU[0], self._accV)
        else:
            self._gamma_side('v', self._Zj, rate_vec=self._sumU[0], sums_arg=self._accV, zero=False)
        self._v_sums_in_acc = True

    # ---- [r6] a2 and U_hat of the cell side are evaluated on access ---------------------------------------------------------
    # Inside a sweep nothing reads them: a2[i, :] = alpha2 + sum_j V_hat is the same K numbers for every cell (gap.py:98) and
    # U_hat = a1 / a2 (gap.py:101) enters the sweep only through its column sums, which the update kernel forms itself.  The
    # kernel therefore writes the K rate values once (_a2_row) and model.a2 / model.U_hat / factors() / state() / save()
    # materialise the (n, K) float64 matrices when somebody asks -- a broadcast and a float64 division, bit for bit what the
    # kernel would have stored.  1.6 of the 4.8 GB of the cell-side update at configs[3] (ORIANA_LAZY_U=0: stored every sweep).
    _u_stale = False
    _U_buf = None
    _a2_row = None
    _lazy_ok = True

    @property
    def _U_hat(self):
        if self._u_stale:
            torch.div(self.a1.tensor, self._a2_row, out=self._U_buf)           # Gamma._mean, gamma.py:37-46
            self._u_stale = False
        return self._U_buf

    @_U_hat.setter
    def _U_hat(self, t):
        self._U_buf, self._u_stale = t, False

    def _u_on_access(self):
        n, K = self.n, self.k
        row = self._a2_row
        self.a2.defer(lambda: row.clone().expand(n, K).contiguous())
        self._u_stale = True

    def step(self):
        FactorModel.step(self)
        if self._graph is not None and self._a2_row is not None and self._lazy_ok:
            self._u_on_access()              # (a replayed graph ran the kernel again: what was materialised is stale)

    def _init_extra(self):
        if os.environ.get('ORIANA_LAZY_U', '1') != '0':
            from ..parameters import LazyParameter
            n, K, dev = self.n, self.k, self.device
            self._a2_row = torch.ones(K, dtype=torch.float64, device=dev)
            self.a2 = LazyParameter((n, K), dev, lambda: torch.ones(n, K, dtype=torch.float64, device=dev))   # gap.py:43 (ones)
        if self.sharded and self._ws.Kp == self.k:
            # the gene-side update of the packed exchange: Z in the caller's gene order (cleared by the factor preparation's
            # launch every sweep) and the all-ones factor of  Z[o] = 1 * Zx[p]
            self._Zj_o = torch.zeros(self.m, self.k, dtype=torch.float32, device=self.device)
            self._ones_v = torch.ones(self.m, self._ws.Kp, dtype=torch.float32, device=self.device)

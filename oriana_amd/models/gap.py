# -*- coding: utf-8 -*-
"""pCMF = Gamma-Poisson factor model (reference oriana/models/gap.py:14-135)."""
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
        self._gamma_side_finalize('u', self._Zi, ws.FU, ws.R, ct.row_perm, self._sumV[0], self._sumU, nslab=ws.row_gene_splits,
                                  slab_row0=ws.row_slab_row0)
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

    def _init_extra(self):
        import torch
        if self.sharded and self._ws.Kp == self.k:
            # the gene-side update of the packed exchange: Z in the caller's gene order (cleared by the factor preparation's
            # launch every sweep) and the all-ones factor of  Z[o] = 1 * Zx[p]
            self._Zj_o = torch.zeros(self.m, self.k, dtype=torch.float32, device=self.device)
            self._ones_v = torch.ones(self.m, self._ws.Kp, dtype=torch.float32, device=self.device)

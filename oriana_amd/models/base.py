# -*- coding: utf-8 -*-
"""Factor models: the reference's driver surface (constructor, ``step()``, ``factors()``, public
attribute names -- reference oriana/models/base.py:13-130) over device-resident state and the HIP
kernels of include/oriana_hip.h.  One process per GPU; cells (rows) are sharded across ranks.

What is kept from the reference: attribute names and shapes (``alpha1 .. p_s`` are ``Parameter``
objects holding float64 buffers, parameters.py:8-32), the order of the updates inside a sweep, the
float32 / float64 split of every quantity, the clamps, and -- behind ``reference_quirks=True`` --
the reference's index quirk (zigap.py:94).  What is not: the dense n x m rate matrix is never
materialised (``UV`` is evaluated on demand), X is packed once instead of re-cast every sweep.
"""
import numpy as np
import torch

from .. import engine
from .._lib import call, ptr, stream_ptr, OrianaHipError
from ..parameters import Parameter
from ..dims import Dimensions
from .. import dist as odist

__all__ = ['FactorModel']


class _Buffer:
    """Minimal stand-in for a reference node (``model.U[:]``, ``.buffer``, ``.asarray()``)."""

    def __init__(self, getter):
        self._getter = getter

    @property
    def buffer(self):
        return self._getter()

    def asarray(self):
        return self._getter().detach().cpu().numpy()

    def __getitem__(self, key):
        return self._getter()[key].detach().cpu().numpy()


def _is_sparse_input(c):
    """SciPy sparse matrices and CountMatrix objects built from one (`._sparse`).  A torch sparse tensor is
    not a supported container: say so instead of failing inside SciPy."""
    import scipy.sparse as sp
    if isinstance(c, torch.Tensor) and (c.is_sparse or c.layout != torch.strided):
        raise TypeError('torch sparse tensors are not supported: pass a SciPy sparse matrix, a dense array / tensor '
                        'or engine.CountTiles')
    return sp.issparse(c) or sp.issparse(getattr(c, '_sparse', None))


class FactorModel:
    """Base class (reference models/base.py:13-56).

    Parameters
    ----------
    cmatrix : object with ``.shape`` and ``.as_array()`` (reference CountMatrix, base.py:22-23,
        gap.py:31), a NumPy array, a torch tensor, or an ``engine.CountTiles`` already resident
        on the device.  With a process group this is the LOCAL row shard.
    k : number of factors.  use_factors : warm-start a1 / b1 from NMF factors (base.py:37-40).
    init : optional ``(a1, b1)`` arrays (post-clamp initial shapes) -- bypasses NMF / host RNG; or
        ``'nmf'`` / ``'random'``: starts computed on the device (models/deviceinit.py), for counts that are
        already resident, sparse or row-sharded (``seed`` keys them by global row).
    device : torch device (default cuda).  process_group : torch.distributed group for row sharding.
    reference_quirks : reproduce zigap.py:94 (``D_hat[i, k]``) -- see SURVEY.md 8(a) policy.
    dense_density : genes expressed in at least this share of the cells are evaluated densely on the bf16 matrix cores in
        float32-equivalent arithmetic (hybrid layout, csrc/dense_pass.hip; DESIGN_HISTORY.md section 10).  Every model takes it:
        inside the ZI models D_hat = 1 at every non-zero count, so their nest is the pCMF nest plus the D_hat[i, k] weight
        on the gene side; the sparse models' den runs against the masked FV image, their accumulation against FV * S_hat,
        their log sums through a second gene-side pass.  ``'auto'`` for the ZI / sparse models: the same threshold, but only
        when the genes above it hold at least 75 % of the non-zeros (i.e. on data that are about half dense or denser).
        ``'auto'``: engine.auto_dense_density -- engine.DENSE_DENSITY_DEFAULT (or the environment's ORIANA_DENSE_DENSITY;
        ``0`` / ``off`` disables) for matrices of at least 2e8 entries and a K the dense kernels are compiled for; ignored
        for a prebuilt ``engine.CountTiles`` (its own layout is used).
    """

    zi = False
    sparse = False
    _graph_capturing = False

    def __init__(self, cmatrix, k=2, use_factors=True, tau=0.5, init=None, device=None, process_group=None,
                 reference_quirks=True, n_total=None, seed=0, dense_density='auto'):
        if not torch.cuda.is_available():
            raise OrianaHipError('oriana_amd needs a ROCm GPU: the CAVI kernels are HIP only (no CPU fallback)')
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.cmatrix = cmatrix
        self.k = int(k)
        self.tau = tau
        self.use_factors = use_factors
        self.reference_quirks = reference_quirks
        self.seed = int(seed)
        self.pg = process_group
        self.world = odist.world_size(process_group)
        self.sharded = odist.sharded(process_group)     # (also the one-rank rehearsal mode, dist._forced)

        X_host = None
        # Under row sharding every rank must pack the genes in the SAME internal order (the replicated
        # gene-side matrices of the on-device NMF start live in packed order): the per-gene counts that
        # define the order are summed over the shards.
        rf = (lambda t: odist.all_reduce_sum(t, process_group)) if self.sharded else None
        dd = self._dense_density(dense_density, cmatrix, n_total, init)
        if isinstance(cmatrix, engine.CountTiles):
            self.counts = cmatrix
        elif _is_sparse_input(cmatrix):
            A = cmatrix._sparse if hasattr(cmatrix, '_sparse') else cmatrix
            X_host = A                     # the host-side initialisation reads it in sparse form
            self.counts = engine.CountTiles.from_scipy(A, self.device, reduce_fn=rf, dense_density=dd, n_total=n_total,
                                                       dense_min_share=getattr(self, '_dense_min_share', 0.0))
        else:
            X = cmatrix.as_array() if hasattr(cmatrix, 'as_array') else cmatrix
            if not isinstance(X, torch.Tensor):
                X = np.asarray(X)
                X_host = X
            self.counts = engine.CountTiles.from_dense(X, self.device, reduce_fn=rf, dense_density=dd, n_total=n_total,
                                                       dense_min_share=getattr(self, '_dense_min_share', 0.0))
        self.n = self.counts.n
        self.m = self.p = self.counts.m
        self.n_total = int(n_total) if n_total is not None else odist.sum_int(self.n, process_group, self.device)
        self.dims = Dimensions({'n': self.n, 'm': self.m, 'p': self.p, 'k': self.k})
        if self.zi and self.reference_quirks and self.k > self.m:
            raise ValueError('reference_quirks=True needs k <= number of genes (zigap.py:94 reads D_hat[i, k])')

        K, n, m, dev = self.k, self.n, self.m, self.device
        f64 = dict(dtype=torch.float64, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        # prior hyper-parameters (build_u_node / build_v_node, gap.py:19-27): every value is
        # overwritten by the first M-step except alpha2 = beta2 = 1, which that M-step reads
        self.alpha1 = Parameter(torch.ones(K, **f64))
        self.alpha2 = Parameter(torch.ones(K, **f64))
        self.beta1 = Parameter(torch.ones(K, **f64))
        self.beta2 = Parameter(torch.ones(K, **f64))
        # variational parameters (define_variational_distribution, gap.py:34-44)
        a1_0, b1_0, nmf = self._initial_shapes(X_host, init)
        self.nmf_factors = nmf
        # (copy=True: a device tensor given as `init` must not become the parameter buffer of the model)
        self.a1 = Parameter(a1_0.to(copy=True, **f64).clamp_(min=1e-15).contiguous())       # gap.py:55
        self.a2 = Parameter(torch.ones(n, K, **f64))
        self.b1 = Parameter(b1_0.to(copy=True, **f64).clamp_(min=1e-15).contiguous())       # gap.py:65
        self.b2 = Parameter(torch.ones(m, K, **f64))
        # expectations (device); exposed as NumPy through the properties below
        self._U_hat = torch.empty(n, K, **f64)
        self._V_hat = torch.empty(m, K, **f64)
        self._log_U_hat = torch.empty(n, K, **f32)
        self._log_V_hat = torch.empty(m, K, **f32)
        self._Zi = torch.empty(max(n, 1), K, **f32)
        # everything a sweep exchanges between row shards lives in ONE packed float32 buffer (dist.SweepExchange):
        # the per-gene sums are views into it, written in place by the kernels
        f32_seg = {'Zj': (max(m, 1), K)}
        if self.sparse:
            f32_seg['Zlog'] = (max(m, 1), K)
        f64_seg = {'DtU': (m, K)} if self.zi else {}
        f64_seg['sumU'] = (2, K)
        self._xch = odist.SweepExchange(dev, process_group, f32_seg, f64_seg)
        self._Zj = self._xch.f32['Zj']
        self._sumU = torch.zeros(2, K, **f64)        # [sum_i U_hat, sum_i log_U_hat]
        self._sumV = torch.zeros(2, K, **f64)
        # the gene side's column sums while a sweep accumulates them (cleared with the other scratch of the sweep by the
        # factor preparation's launch; the M-step copies them to _sumV, which the next cell-side update reads)
        self._accV = torch.zeros(2, K, **f64)
        self._v_sums_in_acc = False
        self._ws = engine.ZWorkspace(self.counts, K, need_sw=False, need_srow=False)     # (s_rs: allocated on first use)
        self._init_extra()
        self.U = _Buffer(lambda: self._U_hat)
        self.V = _Buffer(lambda: self._effective_V())
        self.UV = _Buffer(lambda: self._U_hat @ self._effective_V().t())          # lazy Einsum('nk,mk->nm')
        self.n_sweeps = 0
        self._graph = None
        self.initialize_parameters()

    def _dense_density(self, dense_density, cmatrix, n_total, init=None):
        """The density threshold of the hybrid layout for this model, or None."""
        if isinstance(cmatrix, engine.CountTiles):
            return None
        # 'auto' for the ZI / sparse models: the dense block is neutral for them at the benchmark's 90 % zeros and pays
        # 1.4-2 x at the reference generator's own ~50 % (profiles/r04_zigap_hybrid_ab.json, r04_sparsegap_hybrid_ab.json), so
        # they take it only when the genes above the threshold hold most of the non-zeros (decided at packing, from the
        # per-gene counts: CountTiles.dense_order min_share)
        self._dense_min_share = 0.75 if ((self.zi or self.sparse) and isinstance(dense_density, str)) else 0.0
        if isinstance(init, str) and init == 'nmf':      # the on-device NMF start walks the sliced layout only
            return None
        shape = getattr(cmatrix, 'shape', None)
        rows = int(n_total) if n_total is not None else (int(shape[0]) if shape is not None else 0)
        cols = int(shape[1]) if shape is not None else 0
        if isinstance(dense_density, str):
            if dense_density != 'auto':
                raise ValueError("dense_density must be 'auto', a density in (0, 1] or 0 / None")
            return engine.auto_dense_density(rows, cols, self.k)
        if not dense_density or not engine.dense_supported(self.k):
            return None
        return float(dense_density)

    # ---- initial shapes -------------------------------------------------------------------------
    def _initial_shapes(self, X_host, init):
        """a1, b1 before the clamp.  With ``init`` given: exactly those.  Otherwise the reference's
        host-side sequence is replayed (same np.random calls in the same order, then scikit-learn
        NMF) so that ``np.random.seed(s)`` before the constructor gives the reference's start."""
        if isinstance(init, str):
            # on-device starts (deviceinit.py): no host copy of X, sharding-invariant
            from . import deviceinit
            seed = int(getattr(self, 'seed', 0))
            row0 = deviceinit.global_row_offset(self.n, self.pg, self.device)
            if init == 'nmf':
                W, H = deviceinit.device_nmf(self.counts, self.k, seed=seed, pg=self.pg, row0=row0)
                if self.use_factors:
                    return W, H, (W, H)
                a1, b1 = deviceinit.random_shapes(self.n, self.m, self.k, seed=seed + 1, device=self.device, row0=row0)
                return a1, b1, (W, H)
            if init == 'random':
                a1, b1 = deviceinit.random_shapes(self.n, self.m, self.k, seed=seed, device=self.device, row0=row0)
                return a1, b1, None
            raise ValueError("init must be (a1, b1), 'nmf' or 'random'")
        if init is not None:
            a1, b1 = init
            a1 = a1 if isinstance(a1, torch.Tensor) else torch.as_tensor(np.asarray(a1, dtype=np.float64))
            b1 = b1 if isinstance(b1, torch.Tensor) else torch.as_tensor(np.asarray(b1, dtype=np.float64))
            if tuple(a1.shape) != (self.n, self.k) or tuple(b1.shape) != (self.m, self.k):
                raise ValueError('init shapes must be (n, k) and (m, k)')
            return a1, b1, None
        if X_host is None:
            raise ValueError("init=(a1, b1), 'nmf' or 'random' is required when the count matrix is already on the device")
        if self.world > 1:
            raise ValueError("init=(a1, b1), 'nmf' or 'random' is required under row sharding (the host NMF sees the whole matrix)")
        from .hostinit import reference_initial_shapes
        a1, b1, nmf = reference_initial_shapes(type(self).__name__, X_host, self.k, self.use_factors)
        return torch.from_numpy(a1), torch.from_numpy(b1), nmf

    def _init_extra(self):
        pass

    def _effective_V(self):
        return self._V_hat

    # ---- reference surface ------------------------------------------------------------------------
    def initialize_parameters(self):
        """base.py:43-52: variational initialisation was done in __init__; expectations; M-step."""
        self.update_expectations()
        self.update_prior_hyper_parameters()

    def step(self):
        """One CAVI sweep (base.py:54-56)."""
        if self._graph is not None:
            self._graph.replay()
        else:
            self._sweep()
        self.n_sweeps += 1

    def _sweep(self):
        self.update_variational_parameters()   # E-step
        self.update_prior_hyper_parameters()   # M-step

    def capture_graph(self):
        """Capture one sweep into a hipGraph and replay it from step() on (single process only:
        the launches of a sweep have no host-side decision, so small problems stop being
        launch-bound).  The state tensors are updated in place, exactly as without the graph."""
        if self.sharded:
            raise RuntimeError('graph capture is for single-process models (collectives are not captured)')
        if self.zi:
            # the lazy p_d of the ZI models is host-side state (snapshot + deferred evaluation) that a replayed
            # graph would not refresh: model.p_d / state() would go stale
            raise RuntimeError('graph capture is not available for the zero-inflated models (p_d is evaluated lazily on the host side)')
        if self._graph is not None:
            return self
        # (a captured sweep replays fixed buffers: the FU double buffer of the fused preparation would alternate)
        self._graph_capturing = True
        self._ws.fu_pending = False
        self._sweep()                          # warm-up: allocations, lazy scratch buffers
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._sweep()
        self.n_sweeps += 1                     # the warm-up sweep
        self._graph = g
        return self

    def fit(self, n_iter=50):
        """The loop user scripts write around step() (reference main.py:37-51)."""
        for _ in range(int(n_iter)):
            self.step()
        return self

    def factors(self):
        """base.py:97-98: (U[:], V[:]) as host arrays."""
        return self.U[:], self.V[:]

    # expectations as host NumPy (the reference stores ndarrays in these attributes)
    @property
    def U_hat(self):
        return self._U_hat.cpu().numpy()

    @property
    def V_hat(self):
        return self._V_hat.cpu().numpy()

    @property
    def log_U_hat(self):
        return self._log_U_hat.cpu().numpy()

    @property
    def log_V_hat(self):
        return self._log_V_hat.cpu().numpy()

    # ---- pieces shared by the four models ----------------------------------------------------------
    _ver = 0

    def _touch(self):
        """Counts the writes to the expectations a product kept across sweeps depends on (V_hat, S_hat, D_hat)."""
        self._ver += 1

    def _gamma_side(self, side, Z, zmul=None, rate_vec=None, rate_mat=None, rmul=None, update=True, sums_arg=None, zero=True):
        """One side of update_variational_parameters + Gamma.mean / meanlog (oriana_gamma_update).  `sums_arg`: where the
        column sums go (default: _sumU / _sumV); zero=False: they are zero already."""
        if side == 'v':
            self._touch()
        if side == 'u':
            s1, s2, E, Elog, sums, p1, p2, r = self.a1, self.a2, self._U_hat, self._log_U_hat, self._sumU, self.alpha1, self.alpha2, self.n
        else:
            s1, s2, E, Elog, sums, p1, p2, r = self.b1, self.b2, self._V_hat, self._log_V_hat, self._sumV, self.beta1, self.beta2, self.m
        if sums_arg is not None:
            sums = sums_arg
        if zero:
            sums.zero_()
        prep = self._prep_outputs(side, packed_rows=False)
        call('oriana_gamma_update_prep', ptr(s1.tensor), ptr(s2.tensor), ptr(E), ptr(Elog), ptr(sums[0]), ptr(sums[1]),
             ptr(p1.tensor), ptr(p2.tensor), ptr(Z) if update else None, ptr(zmul), ptr(rate_vec), ptr(rate_mat),
             ptr(rmul), r, self.k, ptr(prep[0]), ptr(prep[1]), ptr(prep[2]), stream_ptr())
        if prep[0] is not None:
            self._ws.fu_pending, self._ws.fu_source = True, Elog.data_ptr()

    def _prep_outputs(self, side, packed_rows):
        """The cell-side update also prepares the next sweep's FU (engine.ZWorkspace.prep_outputs); any update of E[log U]
        invalidates what an earlier one prepared."""
        if side != 'u' or getattr(self, '_ws', None) is None:
            return (None, None, None)
        self._ws.fu_pending = False
        if self._graph_capturing:
            return (None, None, None)
        return self._ws.prep_outputs(packed_rows) or (None, None, None)

    def _gamma_side_finalize(self, side, Z, F, R, row_index, rate_vec, sums, nslab=1, slab_row0=0, a2_row=None):
        """pCMF: Z += F * R (the last step of the responsibility pass, packed rows scattered through `row_index`) and the
        Gamma update of that side in one launch (oriana_gamma_update_finalize).  `sums` (2, K) must be zero on entry.
        `a2_row` (cell side, [r6]): the rate matrix a2 and the mean U_hat are NOT stored -- the K rate values go to a2_row
        (oriana_gamma_update_finalize_lazy); returns True when that form ran, False when the full form did."""
        if side == 'v':
            self._touch()
        prep = self._prep_outputs(side, packed_rows=True)
        if a2_row is not None:
            assert side == 'u'
            from .. import _lib
            rc = _lib.load().oriana_gamma_update_finalize_lazy(
                ptr(self.a1.tensor), ptr(a2_row), ptr(self._log_U_hat), ptr(sums[0]), ptr(sums[1]), ptr(self.alpha1.tensor),
                ptr(self.alpha2.tensor), ptr(Z), ptr(F), ptr(R), int(nslab), int(slab_row0), ptr(row_index), ptr(rate_vec), self.n,
                self.k, ptr(prep[0]), ptr(prep[1]), ptr(prep[2]), stream_ptr())
            if rc == 0:
                if prep[0] is not None:
                    self._ws.fu_pending, self._ws.fu_source = True, self._log_U_hat.data_ptr()
                return True
            if rc != -2:                    # ORIANA_EKRANGE: no vector kernel for this K -- the full form below
                raise OrianaHipError('oriana_gamma_update_finalize_lazy failed with code %d' % rc)
        if side == 'u':
            s1, s2, E, Elog, p1, p2, r = self.a1, self.a2, self._U_hat, self._log_U_hat, self.alpha1, self.alpha2, self.n
        else:
            s1, s2, E, Elog, p1, p2, r = self.b1, self.b2, self._V_hat, self._log_V_hat, self.beta1, self.beta2, self.m
        call('oriana_gamma_update_finalize_prep', ptr(s1.tensor), ptr(s2.tensor), ptr(E), ptr(Elog), ptr(sums[0]), ptr(sums[1]),
             ptr(p1.tensor), ptr(p2.tensor), ptr(Z), ptr(F), ptr(R), int(nslab), int(slab_row0), ptr(row_index), ptr(rate_vec), r,
             self.k, ptr(prep[0]), ptr(prep[1]), ptr(prep[2]), stream_ptr())
        if prep[0] is not None:
            self._ws.fu_pending, self._ws.fu_source = True, Elog.data_ptr()
        return False

    def _mstep_side(self, side):
        if side == 'u':
            call('oriana_mstep_gamma', ptr(self.alpha1.tensor), ptr(self.alpha2.tensor), ptr(self._sumU[0]),
                 ptr(self._sumU[1]), float(self.n_total), self.k, stream_ptr())
        else:
            call('oriana_mstep_gamma', ptr(self.beta1.tensor), ptr(self.beta2.tensor), ptr(self._sumV[0]),
                 ptr(self._sumV[1]), float(self.m), self.k, stream_ptr())

    def update_expectations(self):
        """gap.py:131-135: expectations from the current a1, a2, b1, b2 (no parameter update)."""
        self._gamma_side('u', None, update=False)
        odist.all_reduce_sum(self._sumU, self.pg)
        self._gamma_side('v', None, update=False)
        self._v_sums_in_acc = False

    def update_prior_hyper_parameters(self):
        """gap.py:117-129: both Gamma nodes in one launch."""
        acc = self._v_sums_in_acc
        sv = self._accV if acc else self._sumV
        call('oriana_mstep_gamma_pair', ptr(self.alpha1.tensor), ptr(self.alpha2.tensor), ptr(self._sumU[0]),
             ptr(self._sumU[1]), float(self.n_total), ptr(self.beta1.tensor), ptr(self.beta2.tensor), ptr(sv[0]), ptr(sv[1]),
             float(self.m), ptr(self._sumV) if acc else None, self.k, stream_ptr())
        self._v_sums_in_acc = False

    def update_variational_parameters(self):
        raise NotImplementedError

    def _exchange_start(self, **partials64):
        """First half of the sweep's exchange, called BEFORE the column pass: the float64 partials that already exist
        (the cell-side column sums, D_hat^T U_hat of the ZI models) start their all-reduce asynchronously and the
        column pass runs under it."""
        x = self._xch
        x.put64('sumU', self._sumU)
        for name, t in partials64.items():
            x.put64(name, t)
        x.start64()
        self._xch_names = tuple(partials64)

    def _exchange(self, **partials64):
        """The exchange of a sweep: the float32 buffer (per-gene sums written in place by the column pass) and the
        float64 partials (started by _exchange_start, or handed in here) are summed over the row shards.  Returns the
        reduced float64 partials by name."""
        x = self._xch
        names = getattr(self, '_xch_names', None)
        if names is None:
            x.put64('sumU', self._sumU)
            for name, t in partials64.items():
                x.put64(name, t)
            names = tuple(partials64)
        self._xch_names = None
        x.reduce()
        x.get64('sumU', out=self._sumU)
        return {name: x.get64(name) for name in names}

    # ---- metrics (reference base.py:58-87; loglikelihood_X: sparse_zigap.py:44-51) --------------------
    # The reference defines loglikelihood_X on SparseZIGaP only (the deviances raise AttributeError on
    # the other classes); here the absent nodes read as the constants the formulas reduce to
    # (pi_d = 1, D = 1, S_hat = 1).  With N = {X != 0}, Z = {X == 0}, M = {round(D_hat) == 0} (M never
    # meets N: D_hat = 1 at the non-zeros), Lambda = U_hat (S_hat * V_hat)^T and mu_j = mean_i X_ij:
    #   ll(X | X)      = sum_j nnz_j log pi_j + sum_N (x log x - x)                  (zeros give log 1 = 0)
    #   ll(X | Lambda) = sum_{Z - M} log(pi_j e^-Lambda + 1 - pi_j) + sum_j nnz_j log pi_j
    #                    - sum_N Lambda + sum_N x log Lambda
    #   ll(X | mu)     = sum_j (n - nnz_j) log(pi_j e^-mu_j + 1 - pi_j) + sum_j nnz_j (log pi_j - mu_j)
    #                    + sum_j colsum_j log mu_j
    # The sums over N come from the responsibility kernels (metrics.hip), the sum over Z - M from the
    # f64-MFMA kernel of the D update in metric mode (or, without a D node, from column sums:
    # sum_Z Lambda = sum_k (sum_i U_ik)(sum_j V_jk) - sum_N Lambda).  Float64 semantics: what the
    # reference computes for a float X (for an integer X it truncates every term, sparse_zigap.py:45).
    def _count_constants(self):
        """Per-gene sums / non-zero counts of X and sum_N (x log x - x), sum_N x^2 (all-reduced, cached)."""
        if getattr(self, '_xconst', None) is None:
            dev = self.device
            colsum = torch.zeros(self.m, dtype=torch.float64, device=dev)
            colnnz = torch.zeros(self.m, dtype=torch.float64, device=dev)
            out2 = torch.zeros(2, dtype=torch.float64, device=dev)
            ct = self.counts
            call('oriana_count_stats', ct.sparse_struct, ptr(colsum), ptr(colnnz), ptr(out2), stream_ptr())
            if ct.dense is not None:
                call('oriana_dense_metric', ct.dense.c_struct, None, None, ptr(ct.row_perm), ptr(ct.col_perm), ptr(colsum),
                     ptr(colnnz), ptr(out2), None, self.k, stream_ptr())
            for t in (colsum, colnnz, out2):
                odist.all_reduce_sum(t, self.pg)
            self._xconst = (colsum, colnnz, out2)
        return self._xconst

    def _metric_terms(self):
        ct, K, n, m, dev = self.counts, self.k, self.n, self.m, self.device
        st = stream_ptr()
        U, V = self._U_hat, self._effective_V().contiguous()
        ws = self._ws
        FU, FV = ws.extra('MU', n), ws.extra('MV', m)
        call('oriana_factor_cast_f32', ptr(FU), ptr(U), None, ptr(ct.row_perm), n, K, st)
        call('oriana_factor_cast_f32', ptr(FV), ptr(V), None, ptr(ct.col_perm), m, K, st)
        if ws.s_rs is None:
            ws.s_rs = torch.zeros(max(ct.rslots, 1), dtype=torch.float32, device=dev)
        ws.tile_flag.zero_()
        # with these factors the row pass leaves s = x / Lambda in the row-side slots
        # (hybrid layout: the sliced part covers the packed genes [gd, m); the dense genes are summed in float64)
        call('oriana_row_pass', ct.sparse_struct, ptr(FU), ptr(FV) + 4 * ct.gd * ws.Kp, None, ptr(ws.R), ptr(ws.s_cs), None,
             ptr(ws.s_rs), ptr(ws.tile_flag), K, st)
        nz = torch.zeros(4, dtype=torch.float64, device=dev)      # sum Lambda, sum x log Lambda, sum Lambda^2, sum x Lambda
        call('oriana_metric_nnz', ct.sparse_struct, ptr(ws.s_rs), ptr(U), ptr(V), K, ptr(nz), st)
        if ct.dense is not None:
            call('oriana_dense_metric', ct.dense.c_struct, ptr(U), ptr(V), ptr(ct.row_perm), ptr(ct.col_perm), None, None, None,
                 ptr(nz), K, st)
        zz = torch.zeros(2, dtype=torch.float64, device=dev)      # over Z - M: sum log(pi e^-Lambda + 1 - pi), sum Lambda^2
        if self.zi:
            # (the gene axis of the dropout node's matrices is padded with inert genes to a multiple of 4, zigap.py _init_zi:
            # pi_d = 0 there, so log(pi e^-Lambda + 1 - pi) = 0, and Lambda = 0)
            call('oriana_dropout_metric', ptr(zz), ptr(self._Dp), ptr(U), ptr(self._padG(V, 'Vm')), ptr(self._padG(self.pi_d.tensor, 'pim')),
                 ptr(self._nzmask), n, self._mp, K, st)
            odist.all_reduce_sum(zz, self.pg)
            odist.all_reduce_sum(nz, self.pg)
        else:
            su = U.sum(0)
            gram_u = U.t() @ U
            packed = torch.cat([nz, su, gram_u.reshape(-1)])
            odist.all_reduce_sum(packed, self.pg)
            nz, su, gram_u = packed[:4], packed[4:4 + K], packed[4 + K:].reshape(K, K)
            all_lam = (su * V.sum(0)).sum()
            all_lam2 = (gram_u * (V.t() @ V)).sum()
            zz[0] = -(all_lam - nz[0])
            zz[1] = all_lam2 - nz[2]
        return nz, zz

    def _log_pi(self):
        if self.zi:
            return torch.log(self.pi_d.tensor), self.pi_d.tensor
        one = torch.ones(self.m, dtype=torch.float64, device=self.device)
        return torch.zeros_like(one), one

    def _loglikelihoods(self):
        colsum, colnnz, xc = self._count_constants()
        nz, zz = self._metric_terms()
        lpi, pi = self._log_pi()
        has = colnnz > 0
        nnz_lpi = torch.where(has, colnnz * lpi, torch.zeros_like(lpi)).sum()
        ll_x = nnz_lpi + xc[0]
        ll_uv = zz[0] + nnz_lpi - nz[0] + nz[1]
        ntot = float(self.n_total)
        mu = colsum / ntot
        zero_term = (ntot - colnnz) * torch.log(pi * torch.exp(-mu) + (1.0 - pi))
        nz_term = torch.where(has, colnnz * (lpi - mu) + colsum * torch.log(torch.where(has, mu, torch.ones_like(mu))),
                              torch.zeros_like(mu))
        ll_mean = (zero_term + nz_term).sum()
        return float(ll_x), float(ll_uv), float(ll_mean), nz, zz, xc

    def loglikelihood_X(self, given='factors'):
        """The zero-inflated Poisson log-likelihood of the counts (reference sparse_zigap.py:44-51) for the three Poisson
        means the reference's metrics put into the UV node before calling it: 'factors' -- Lambda = U_hat (S_hat * V_hat)^T,
        zeroed where round(D_hat) == 0 (base.py:64-68: the buffer reconstruction_deviance leaves behind, i.e. what a call
        after it returns); 'counts' -- Lambda = X (base.py:59-63); 'mean' -- Lambda = the per-gene mean (base.py:76-77)."""
        ll_x, ll_uv, ll_mean, _, _, _ = self._loglikelihoods()
        try:
            return {'factors': ll_uv, 'counts': ll_x, 'mean': ll_mean}[given]
        except KeyError:
            raise ValueError("given must be 'factors', 'counts' or 'mean'") from None

    def reconstruction_deviance(self):
        """-2 (ll(X | U_hat V_hat^T) - ll(X | X)), reference base.py:58-69."""
        ll_x, ll_uv, _, _, _, _ = self._loglikelihoods()
        return -2.0 * (ll_uv - ll_x)

    def explained_deviance(self):
        """(ll(X | U_hat V_hat^T) - ll(X | mean)) / (ll(X | X) - ll(X | mean)), reference base.py:71-82
        evaluated on the current expectations (the reference reads the node buffers its own
        reconstruction_deviance call left behind, experiments/clustering.py:26-27)."""
        ll_x, ll_uv, ll_mean, _, _, _ = self._loglikelihoods()
        return (ll_uv - ll_mean) / (ll_x - ll_mean)

    def frobenius_norm(self):
        """|| Lambda - X ||_F with Lambda = U_hat V_hat^T, zeroed where round(D_hat) == 0 (reference
        base.py:84-87 on the UV buffer the deviance calls leave behind)."""
        _, _, _, nz, zz, xc = self._loglikelihoods()
        # sum_N (Lambda - x)^2 + sum_{Z - M} Lambda^2
        return float(torch.sqrt(torch.clamp(nz[2] - 2.0 * nz[3] + xc[1] + zz[1], min=0.0)))

    # ---- state I/O (tests, checkpoints) ---------------------------------------------------------------
    _PARAMS = ('alpha1', 'alpha2', 'beta1', 'beta2', 'a1', 'a2', 'b1', 'b2', 'pi_d', 'p_d', 'pi_s', 'p_s')

    def state(self):
        """Host copy of every parameter and expectation, keyed like the oracle / golden files."""
        out = {}
        for k in self._PARAMS:
            if hasattr(self, k):
                out[k] = getattr(self, k).asarray()
        out.update(U_hat=self.U_hat, V_hat=self.V_hat, log_U_hat=self.log_U_hat, log_V_hat=self.log_V_hat)
        if hasattr(self, '_S_hat'):
            out['S_hat'] = self.S_hat
        return out

    def load_state(self, st):
        """Overwrite parameters (and recompute nothing): used to start a sweep from a given state."""
        for k in self._PARAMS:
            if k in st and hasattr(self, k):
                getattr(self, k).tensor.copy_(torch.as_tensor(np.asarray(st[k], dtype=np.float64)).to(self.device))
        for k, t in (('U_hat', self._U_hat), ('V_hat', self._V_hat), ('log_U_hat', self._log_U_hat),
                     ('log_V_hat', self._log_V_hat)):
            if k in st:
                t.copy_(torch.as_tensor(np.asarray(st[k])).to(self.device, dtype=t.dtype))
        self._touch()
        self._ws.fu_pending = False             # (E[log U] was overwritten)
        # column sums that the next sweep reads
        self._sumU[0] = self._U_hat.sum(0); self._sumU[1] = self._log_U_hat.double().sum(0)
        odist.all_reduce_sum(self._sumU, self.pg)
        self._sumV[0] = self._V_hat.sum(0); self._sumV[1] = self._log_V_hat.double().sum(0)
        self._load_extra(st)

    def _load_extra(self, st):
        pass

    def save(self, path):
        """Checkpoint: the state of this rank (state(), plus the sweep count and the shapes) as one .npz file.  The
        reference has no save / load; under row sharding every rank saves its own rows."""
        st = self.state()
        st['meta/n_sweeps'] = np.int64(self.n_sweeps)
        st['meta/shape'] = np.asarray([self.n, self.m, self.k], dtype=np.int64)
        st['meta/model'] = np.asarray(type(self).__name__)
        np.savez(path, **st)

    def restore(self, path):
        """Load a checkpoint written by save() into this model (same class, same counts and k): the next step()
        continues the run (the expectations are stored, nothing is recomputed)."""
        with np.load(path, allow_pickle=False) as f:
            shape = tuple(int(v) for v in f['meta/shape'])
            if shape != (self.n, self.m, self.k) or str(f['meta/model']) != type(self).__name__:
                raise ValueError('checkpoint of %s %s does not fit %s %s' % (str(f['meta/model']), shape, type(self).__name__,
                                                                              (self.n, self.m, self.k)))
            self.load_state({k: f[k] for k in f.files if not k.startswith('meta/')})
            self.n_sweeps = int(f['meta/n_sweeps'])
        return self

# -*- coding: utf-8 -*-
"""Initial shapes computed on the device.

The reference warm-starts ``a1``, ``b1`` from scikit-learn's NMF of the whole dense matrix on the host
(reference models/base.py:38-40, gap.py:49-50, 59-60) -- ``hostinit.py`` replays exactly that for
seeded parity.  When the counts are already resident on the GPU, sparse, or row-sharded, that route
does not exist; this module provides the same kind of start without leaving the device:

* ``device_nmf``: non-negative factorisation ``X ~ W H^T`` (Frobenius loss) by multiplicative updates
  (Lee & Seung), whose two large products are the streaming passes of the CAVI kernels themselves --
  ``X H`` is ``oriana_row_spmm`` and ``X^T W`` is ``oriana_col_pass`` with the counts as scalars -- plus
  K x K Gram matrices.  It is a stand-in for an UNPINNED third-party routine (SURVEY 8c treats the NMF
  output as an input fixture): it is pinned against a NumPy restatement of the same multiplicative updates
  from the same start (oracle/nmf_oracle.py, tests/test_models_gpu.py) and by its properties -- non-negative
  factors, a loss that never increases, sharding invariance.
* ``random_shapes``: the ``use_factors=False`` start, ``Gamma(1)`` draws (gap.py:52, 62), from a
  device generator keyed by the GLOBAL row index, so that every sharding starts from the same state.
"""
import torch

from .. import dist as odist
from .. import engine
from .._lib import call, ptr, stream_ptr

__all__ = ['device_nmf', 'random_shapes', 'global_row_offset']

_BLOCK = 4096


def global_row_offset(n_local, pg, device):
    """First global row of this rank's shard (exclusive scan of the shard sizes)."""
    world = odist.world_size(pg)
    if world == 1:
        return 0
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    sizes[odist.rank(pg)] = int(n_local)
    odist.all_reduce_sum(sizes, pg)
    return int(sizes[:odist.rank(pg)].sum().item())


def _rows_by_global_block(n, K, row0, seed, device, draw):
    """(n, K) float64 matrix whose row r depends only on (seed, row0 + r): blocks of _BLOCK global rows
    are drawn from their own generator state."""
    out = torch.empty(n, K, dtype=torch.float64, device=device)
    g = torch.Generator(device=device)
    r = 0
    while r < n:
        blk = (row0 + r) // _BLOCK
        lo = blk * _BLOCK
        g.manual_seed(int(seed) * 1000003 + blk)
        full = draw((_BLOCK, K), g)
        a = row0 + r - lo
        take = min(_BLOCK - a, n - r)
        out[r:r + take] = full[a:a + take]
        r += take
    return out


def _gamma1(shape, g):
    # Gamma(shape 1, scale 1) = Exponential(1): -log(U)
    u = torch.rand(shape, generator=g, device=g.device, dtype=torch.float64)
    return -torch.log1p(-u)


def random_shapes(n, m, K, seed=0, device='cuda', row0=0):
    """a1 ~ Gamma(1) (n, K) keyed by global rows, b1 ~ Gamma(1) (m, K) identical on every rank."""
    device = torch.device(device)
    a1 = _rows_by_global_block(n, K, row0, seed, device, _gamma1)
    b1 = _rows_by_global_block(m, K, 0, seed + 7919, device, _gamma1)
    return a1, b1


def device_nmf(counts, K, n_iter=100, tol=1e-4, seed=0, pg=None, row0=None, return_loss=False, init=None):
    """W (n, K), H (m, K) float64 >= 0 with X ~ W H^T, by multiplicative updates on the packed tiles.

    ``counts``: engine.CountTiles of this rank's rows; under row sharding H and the K x K Gram matrices
    are all-reduced, W stays local.  Stops after ``n_iter`` sweeps or when the loss improves by less than
    ``tol`` relative to the first sweep's improvement.  Deterministic for a given (seed, global shape).
    ``init``: optional (W0 (n, K), H0 (m, K)) start instead of the seeded one (oracle comparisons)."""
    ct = counts
    n, m, dev = ct.n, ct.m, ct.device
    Kp = engine.kpad(K)
    st = stream_ptr()
    f32 = dict(dtype=torch.float32, device=dev)
    if row0 is None:
        row0 = global_row_offset(n, pg, dev)
    n_total = odist.sum_int(n, pg, dev)

    # the counts as per-slot scalars in both slot orders: with unit factors the row pass's
    # s = x / <e0, e0> is x itself
    e_u = torch.zeros(max(n, 1), Kp, **f32); e_u[:, 0] = 1.0
    e_v = torch.zeros(max(m, 1), Kp, **f32); e_v[:, 0] = 1.0
    x_cs = torch.zeros(max(ct.cslots, 1), **f32)
    x_rs = torch.zeros(max(ct.rslots, 1), **f32)
    R = torch.zeros(max(n, 1), Kp, **f32)
    C = torch.zeros(max(m, 1), Kp, **f32)
    flag = torch.zeros(max(ct.nrb * ct.ncb, 1), dtype=torch.int32, device=dev)
    call('oriana_row_pass', ct.c_struct, ptr(e_u), ptr(e_v), None, ptr(R), ptr(x_cs), None, ptr(x_rs), ptr(flag), K, st)
    del e_u, e_v

    # constants of X: sum x (for the scale of the start) and sum x^2 (for the loss)
    colsum = torch.zeros(m, dtype=torch.float64, device=dev)
    colnnz = torch.zeros(m, dtype=torch.float64, device=dev)
    xc = torch.zeros(2, dtype=torch.float64, device=dev)
    call('oriana_count_stats', ct.c_struct, ptr(colsum), ptr(colnnz), ptr(xc), st)
    tot = torch.stack([colsum.sum(), xc[1]])
    odist.all_reduce_sum(tot, pg)
    mean_x = float(tot[0]) / max(1.0, float(n_total) * m)
    sum_x2 = float(tot[1])

    # start: scikit-learn's 'random' recipe, sqrt(mean(X) / K) |N(0, 1)|, keyed by global rows
    scale = (mean_x / K) ** 0.5
    absn = lambda shape, g: torch.randn(shape, generator=g, device=g.device, dtype=torch.float64).abs()
    if init is not None:
        W0 = torch.as_tensor(init[0], dtype=torch.float64).to(dev)
        H0 = torch.as_tensor(init[1], dtype=torch.float64).to(dev)
        assert tuple(W0.shape) == (n, K) and tuple(H0.shape) == (m, K)
    else:
        W0 = _rows_by_global_block(n, K, row0, seed, dev, absn) * scale
        H0 = _rows_by_global_block(m, K, 0, seed + 7919, dev, absn) * scale
    rp = ct.row_perm.long() if ct.row_perm is not None else None
    cp = ct.col_perm.long() if ct.col_perm is not None else None
    Wp = torch.zeros(max(n, 1), Kp, **f32)
    Hp = torch.zeros(max(m, 1), Kp, **f32)
    Wp[:n, :K] = (W0[rp] if rp is not None else W0).float()
    Hp[:m, :K] = (H0[cp] if cp is not None else H0).float()
    del W0, H0

    eps = 1e-12
    losses = []
    first_gain = None
    for it in range(int(n_iter)):
        # W <- W * (X H) / (W H^T H)
        call('oriana_row_spmm', ct.c_struct, ptr(x_rs), None, ptr(Hp), ptr(R), K, st)
        hth = Hp.t().double() @ Hp.double()
        Wp.mul_(R / (Wp.double() @ hth).float().clamp_min_(eps))
        # H <- H * (X^T W) / (H W^T W)        (sums over the row shards)
        C.zero_()
        engine.col_pass(ct, x_cs, Wp, C, K)
        wtw = Wp.t().double() @ Wp.double()
        odist.all_reduce_sum(C, pg)
        odist.all_reduce_sum(wtw, pg)
        Hp.mul_(C / (Hp.double() @ wtw).float().clamp_min_(eps))
        # loss ||X - W H^T||_F^2 = sum x^2 - 2 <W, X H> + <W^T W, H^T H>, from one more row product
        call('oriana_row_spmm', ct.c_struct, ptr(x_rs), None, ptr(Hp), ptr(R), K, st)
        parts = torch.stack([(Wp.double() * R.double()).sum()])
        odist.all_reduce_sum(parts, pg)
        hth = Hp.t().double() @ Hp.double()
        loss = sum_x2 - 2.0 * float(parts[0]) + float((wtw * hth).sum())
        losses.append(loss)
        if len(losses) >= 2:
            gain = losses[-2] - losses[-1]
            if first_gain is None:
                first_gain = max(gain, 0.0)
            elif gain <= tol * max(first_gain, 1e-300):
                break
    W = torch.empty(n, K, dtype=torch.float64, device=dev)
    H = torch.empty(m, K, dtype=torch.float64, device=dev)
    if rp is not None:
        W[rp] = Wp[:n, :K].double()
    else:
        W.copy_(Wp[:n, :K])
    if cp is not None:
        H[cp] = Hp[:m, :K].double()
    else:
        H.copy_(Hp[:m, :K])
    return (W, H, losses) if return_loss else (W, H)

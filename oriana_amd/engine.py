# -*- coding: utf-8 -*-
"""Host side of the responsibility pass: the resident tiled count matrix and thin wrappers over the
C ABI (include/oriana_hip.h).  torch is used for device memory and streams only."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import OrianaCounts, call, ptr, stream_ptr

TILE = 256
_XDTYPE = {torch.float32: 0, torch.int64: 1, torch.int32: 2, torch.float64: 3}


def kpad(K):
    kp = int(_lib.load().oriana_kpad(int(K)))
    if kp == 0:
        raise _lib.OrianaHipError('K=%d is outside the compiled range (1..256)' % K)
    return kp


def _as_device_chunk(X, r0, r1, device):
    """Rows [r0, r1) of X as a contiguous device tensor of a dtype the packer reads."""
    if isinstance(X, torch.Tensor):
        c = X[r0:r1]
    else:
        c = torch.from_numpy(np.ascontiguousarray(X[r0:r1]))
    if c.dtype not in _XDTYPE:
        c = c.to(torch.float32)
    return c.to(device, non_blocking=False).contiguous()


class CountTiles:
    """The count matrix X of one row shard, resident in HBM as 256 x 256 tiles of non-zero records
    (struct oriana_counts).  Built once: X is constant across sweeps (reference gap.py:29-32)."""

    def __init__(self, n, m, device):
        self.n, self.m = int(n), int(m)
        self.device = torch.device(device)
        self.nrb = (self.n + TILE - 1) // TILE
        self.ncb = (self.m + TILE - 1) // TILE
        nt = self.nrb * self.ncb
        self.tile_cnt = torch.zeros(max(nt, 1), dtype=torch.int32, device=self.device)
        self.row_ptr = torch.zeros(max(nt, 1) * (TILE + 1), dtype=torch.int32, device=self.device)
        self.col_ptr = torch.zeros(max(nt, 1) * (TILE + 1), dtype=torch.int32, device=self.device)
        self.tile_off = None
        self.rowrec = None
        self.ridx = None
        self.nnz = 0
        self._struct = None

    # ---- building -------------------------------------------------------------------------
    def count_chunk(self, chunk, r0):
        assert r0 % TILE == 0 and chunk.is_contiguous() and chunk.shape[1] == self.m
        call('oriana_pack_count', ptr(chunk), _XDTYPE[chunk.dtype], chunk.shape[0], self.m, chunk.stride(0),
             r0 // TILE, self.ncb, ptr(self.tile_cnt), ptr(self.row_ptr), ptr(self.col_ptr), stream_ptr())

    def finish_count(self):
        nt = self.nrb * self.ncb
        off = torch.zeros(nt + 1, dtype=torch.int64, device=self.device)
        if nt:
            off[1:] = torch.cumsum(self.tile_cnt[:nt].to(torch.int64), dim=0)
        self.tile_off = off
        self.nnz = int(off[-1].item())
        self.rowrec = torch.empty(max(self.nnz, 1), dtype=torch.int64, device=self.device)   # 8-byte records
        self.ridx = torch.empty(max(self.nnz, 1), dtype=torch.uint8, device=self.device)

    def fill_chunk(self, chunk, r0, side=None, side_nz=None):
        assert r0 % TILE == 0 and chunk.is_contiguous()
        call('oriana_pack_fill', ptr(chunk), _XDTYPE[chunk.dtype], chunk.shape[0], self.m, chunk.stride(0),
             r0 // TILE, self.ncb, ptr(self.tile_off), ptr(self.row_ptr), ptr(self.col_ptr), ptr(self.rowrec),
             ptr(self.ridx), ptr(side), side.stride(0) if side is not None else 0, ptr(side_nz), stream_ptr())

    def finish(self):
        self._struct = OrianaCounts(self.n, self.m, self.nrb, self.ncb, self.nnz, ptr(self.tile_off),
                                    ptr(self.row_ptr), ptr(self.col_ptr), ptr(self.rowrec), ptr(self.ridx))
        return self

    @classmethod
    def from_dense(cls, X, device='cuda', chunk_bytes=1 << 30, side=None):
        """Pack a dense (n, m) matrix (NumPy or torch, host or device).  `side`: optional dense
        (n, m) float32 DEVICE matrix gathered at the non-zeros (returned as .side_nz)."""
        n, m = X.shape
        self = cls(n, m, device)
        if n == 0 or m == 0:
            self.finish_count()
            self.side_nz = None
            return self.finish()
        rows = max(TILE, (chunk_bytes // max(1, m * 8)) // TILE * TILE)
        for r0 in range(0, n, rows):
            self.count_chunk(_as_device_chunk(X, r0, min(n, r0 + rows), self.device), r0)
        self.finish_count()
        self.side_nz = None
        if side is not None:
            self.side_nz = torch.empty(max(self.nnz, 1), dtype=torch.float32, device=self.device)
        for r0 in range(0, n, rows):
            r1 = min(n, r0 + rows)
            self.fill_chunk(_as_device_chunk(X, r0, r1, self.device), r0,
                            side[r0:r1] if side is not None else None, self.side_nz)
        return self.finish()

    @classmethod
    def from_chunks(cls, n, m, chunk_fn, chunk_rows, device='cuda'):
        """Two passes over `chunk_fn(r0, r1) -> dense device tensor` (deterministic generator)."""
        assert chunk_rows % TILE == 0
        self = cls(n, m, device)
        for r0 in range(0, n, chunk_rows):
            self.count_chunk(chunk_fn(r0, min(n, r0 + chunk_rows)).contiguous(), r0)
        self.finish_count()
        for r0 in range(0, n, chunk_rows):
            self.fill_chunk(chunk_fn(r0, min(n, r0 + chunk_rows)).contiguous(), r0)
        self.side_nz = None
        return self.finish()

    @property
    def c_struct(self):
        return ctypes.byref(self._struct)

    def bytes_resident(self):
        return (self.nnz * 9 + (self.nrb * self.ncb) * (2 * (TILE + 1) * 4 + 12))

    # ---- debugging / tests ------------------------------------------------------------------
    def to_dense(self):
        """Rebuild the dense float32 matrix on the host (tests only)."""
        X = np.zeros((self.n, self.m), dtype=np.float32)
        if self.nnz == 0:
            return X
        off = self.tile_off.cpu().numpy()
        rp = self.row_ptr.cpu().numpy().view(np.uint32).reshape(-1, TILE + 1)
        rec = self.rowrec[:self.nnz].cpu().numpy().view(np.dtype([('x', '<f4'), ('cpos', '<u2'), ('col', 'u1'), ('pad', 'u1')]))
        for rb in range(self.nrb):
            for cb in range(self.ncb):
                t = rb * self.ncb + cb
                for r in range(TILE):
                    a, b = off[t] + rp[t, r], off[t] + rp[t, r + 1]
                    if b > a:
                        X[rb * TILE + r, cb * TILE + rec['col'][a:b].astype(np.int64)] = rec['x'][a:b]
        return X


class ZWorkspace:
    """Scratch buffers of the responsibility pass for one (CountTiles, K)."""

    def __init__(self, ct, K, need_sw=False, need_srow=False):
        self.ct, self.K, self.Kp = ct, int(K), kpad(K)
        dev = ct.device
        f32 = dict(dtype=torch.float32, device=dev)
        nnz1 = max(ct.nnz, 1)
        self.FU = torch.zeros(max(ct.n, 1), self.Kp, **f32)
        self.FV = torch.zeros(max(ct.m, 1), self.Kp, **f32)
        self.R = torch.zeros(max(ct.n, 1), self.Kp, **f32)
        self.C = torch.zeros(max(ct.m, 1), self.Kp, **f32)
        self.s_col = torch.empty(nnz1, **f32)
        self.sw_col = torch.empty(nnz1, **f32) if need_sw else None
        self.s_row = torch.empty(nnz1, **f32) if need_srow else None
        self.tile_flag = torch.zeros(max(ct.nrb * ct.ncb, 1), dtype=torch.int32, device=dev)
        self.timer = None      # set to a KernelTimer to time the launches of a sweep


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are launched on (torch's
    current stream, which is the stream handed to the C ABI)."""

    def __init__(self):
        self.records = []

    def span(self, name):
        return _Span(self, name)

    def summary(self):
        """{name: (count, mean_ms)} -- call after torch.cuda.synchronize()."""
        acc = {}
        for name, a, b in self.records:
            c, t = acc.get(name, (0, 0.0))
            acc[name] = (c + 1, t + a.elapsed_time(b))
        return {k: (c, t / c) for k, (c, t) in acc.items()}

    def reset(self):
        self.records = []


class _Span:
    def __init__(self, timer, name):
        self.timer, self.name = timer, name

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.timer.records.append((self.name, self.a, self.b))


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _span(ws, name):
    t = getattr(ws, 'timer', None)
    return t.span(name) if t is not None else _NoSpan()


def factor_prep(F, logF, mask=None, mu=None):
    r, K = logF.shape
    assert logF.dtype == torch.float32 and logF.is_contiguous()
    call('oriana_factor_prep', ptr(F), ptr(mu), ptr(logF), ptr(mask), r, K, stream_ptr())
    return F


def zq_gap(ws, Z_hat_i, Z_hat_j, log_U_hat, log_V_hat):
    """GaP.compute_Z_q_expectations (reference gap.py:67-80) on the resident tiles: outputs first,
    zero-filled by the callee, returns None."""
    ct, K = ws.ct, ws.K
    _check_f32(Z_hat_i, (ct.n, K)); _check_f32(Z_hat_j, (ct.m, K))
    _check_f32(log_U_hat, (ct.n, K)); _check_f32(log_V_hat, (ct.m, K))
    st = stream_ptr()
    factor_prep(ws.FU, log_U_hat)
    factor_prep(ws.FV, log_V_hat)
    Z_hat_i.zero_(); Z_hat_j.zero_(); ws.C.zero_(); ws.tile_flag.zero_()
    with _span(ws, 'row_pass'):
        call('oriana_row_pass', ct.c_struct, ptr(ws.FU), ptr(ws.FV), None, None, ptr(ws.R), ptr(ws.s_col), None, None,
             ptr(ws.tile_flag), K, st)
    with _span(ws, 'fixup'):
        call('oriana_fixup', ct.c_struct, ptr(ws.tile_flag), ptr(ws.s_col), None, None, ptr(log_U_hat), ptr(log_V_hat),
             None, None, None, None, ptr(Z_hat_i), ptr(Z_hat_j), None, K, 0, st)
    with _span(ws, 'col_pass'):
        call('oriana_col_pass', ct.c_struct, ptr(ws.s_col), ptr(ws.FU), ptr(ws.C), K, st)
    call('oriana_finalize', ptr(Z_hat_i), ptr(ws.FU), ptr(ws.R), None, ct.n, K, 1, st)
    call('oriana_finalize', ptr(Z_hat_j), ptr(ws.FV), ptr(ws.C), None, ct.m, K, 1, st)


def zq_gap_stateless(Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X):
    """oriana_zq_gap_f32: the reference signature on dense device tensors (packs X per call)."""
    for t in (Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() != 2 or not t.is_contiguous():
            raise TypeError('expected 2-D C-contiguous float32 device tensors')
    n, K = log_U_hat.shape
    m = log_V_hat.shape[0]
    _check_f32(Z_hat_i, (n, K)); _check_f32(Z_hat_j, (m, K)); _check_f32(log_V_hat, (m, K)); _check_f32(X, (n, m))
    kpad(K)
    nnz = int(torch.count_nonzero(X).item()) if X.numel() else 0
    nbytes = int(_lib.load().oriana_zq_workspace_bytes(n, m, K, nnz + 64))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=X.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    call('oriana_zq_gap_f32', ptr(Z_hat_i), ptr(Z_hat_j), ptr(log_U_hat), ptr(log_V_hat), ptr(X), n, m, K,
         base, nbytes, stream_ptr())


def _check_f32(t, shape):
    """numba's explicit signature raises TypeError on dtype / ndim mismatch (gap.py:67)."""
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError('expected a 2-D float32 device tensor')
    if not t.is_contiguous():
        raise TypeError('expected a C-contiguous tensor')
    if tuple(t.shape) != tuple(shape):
        raise ValueError('shape mismatch: %s vs %s' % (tuple(t.shape), tuple(shape)))

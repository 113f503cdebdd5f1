# -*- coding: utf-8 -*-
"""Host side of the responsibility pass: the resident tiled count matrix and thin wrappers over the
C ABI (include/oriana_hip.h).  torch is used for device memory and streams only."""
import ctypes

import os

import numpy as np
import torch

from . import _lib
from ._lib import OrianaCounts, OrianaDense, call, ptr, stream_ptr

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
    """The count matrix X of one row shard, resident in HBM in the tiled, sliced non-zero layout
    (struct oriana_counts).  Built once: X is constant across sweeps (reference gap.py:29-32)."""

    def __init__(self, n, m, device, gd=0):
        self.n, self.m = int(n), int(m)
        self.device = torch.device(device)
        # hybrid layout: the first gd packed genes (the densest, a multiple of 32) live in a dense uint16 block
        # (self.dense, csrc/dense_pass.hip); the sliced non-zero layout below then covers the packed genes [gd, m)
        self.gd = int(gd)
        assert self.gd % 32 == 0 and 0 <= self.gd <= self.m
        self.ms = self.m - self.gd
        self.dense = DenseBlock(self.n, self.gd, self.device) if self.gd else None
        self.dense_density = None      # the threshold the layout was built with (from_dense / from_chunks)
        self.nrb = (self.n + TILE - 1) // TILE
        self.ncb = (self.ms + TILE - 1) // TILE
        nt = max(self.nrb * self.ncb, 1)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.tile_nnz = torch.zeros(nt, **i32)
        self.tile_rslots = torch.zeros(nt, **i32)
        self.tile_cslots = torch.zeros(nt, **i32)
        self.rslice = torch.zeros(nt * 17, **i32)
        self.cslice = torch.zeros(nt * 17, **i32)
        self.roff = self.coff = None
        self.rowrec = self.ridx = None
        self.nnz = self.nnz_sparse = self.rslots = self.cslots = 0
        self.col_perm = None      # int32 [m]: packed column c holds gene col_perm[c] (None = identity)
        self.row_perm = None      # int32 [n]: packed row r holds cell row_perm[r] (None = identity)
        self.sort_rows = False
        self.side_nz = None
        self._col_work = {}
        self._struct = None

    # ---- building -------------------------------------------------------------------------
    def set_col_order(self, col_nnz):
        """Pack genes in decreasing order of their non-zero count (`col_nnz`: int64 [m], already
        summed over all row shards): tiles then hold columns of similar density, which shortens
        the padding of the slices.  Internal only -- every dense input / output of the API stays
        in the caller's gene order."""
        order, _ = self.dense_order(col_nnz, 0, None, 0.0)
        self.col_perm = order.to(self.device).to(torch.int32).contiguous()

    @staticmethod
    def dense_order(col_nnz, n_total, bad, density, min_share=0.0):
        """(gene order, gd) of a hybrid layout: genes whose share of non-zero cells is >= `density` and whose counts
        are all integers in [0, 65535) (`bad`: per-gene number of entries that are not) come first, in decreasing
        order of their non-zero count, cut to a multiple of 32; the rest follows in decreasing order.  `min_share`: no
        dense block at all unless those genes hold at least this share of the non-zeros (the ZI / sparse models'
        'auto': the block pays for them from about half-dense data on).  The decision itself is
        oriana_plan_gene_order (csrc/resident.hip: host code of the C ABI, shared with the resident handle)."""
        dev = col_nnz.device
        cn = np.ascontiguousarray(col_nnz.detach().cpu().numpy().astype(np.int64))
        bd = np.ascontiguousarray(bad.detach().cpu().numpy().astype(np.int64)) if bad is not None else None
        m = int(cn.shape[0])
        order = np.empty(max(m, 1), dtype=np.int32)
        gd = ctypes.c_int64(0)
        call('oriana_plan_gene_order', cn.ctypes.data, bd.ctypes.data if bd is not None else None, m, int(n_total),
             float(density or 0.0), float(min_share or 0.0), order.ctypes.data, ctypes.addressof(gd))
        return torch.from_numpy(order[:m].astype(np.int64)).to(dev), int(gd.value)

    def _permute(self, chunk, r0=None, learn=False):
        """Apply the internal orderings to a dense row chunk: genes by col_perm; cells, inside the chunk,
        by decreasing non-zero count (sort_rows, off by default): a 16-row slice advances at the pace of
        its longest row, so rows of similar depth should share a slice.  On cells whose depth varies 6-fold
        this cuts the row-side slots by 24 % (slot efficiency 0.51 -> 0.67, tools/row_order_effect.py) but
        not the time of the row pass, which at that sparsity is dominated by the per-tile costs -- it is a
        memory option.  The chunk's order is learnt on the counting pass and replayed on the fill pass."""
        if self.col_perm is not None:
            chunk = chunk.index_select(1, self.col_perm.to(torch.int64))
        if self.sort_rows and r0 is not None and chunk.shape[0] > 0:
            r1 = r0 + chunk.shape[0]
            if learn:
                if self.row_perm is None:
                    self.row_perm = torch.arange(self.n, dtype=torch.int32, device=self.device)
                rn = (chunk != 0).sum(1)
                order = torch.argsort(rn, descending=True, stable=True)
                self.row_perm[r0:r1] = (order + r0).to(torch.int32)
            else:
                order = self.row_perm[r0:r1].to(torch.int64) - r0
            chunk = chunk.index_select(0, order)
        return chunk.contiguous()

    def count_chunk(self, chunk, r0):
        chunk = self._permute(chunk, r0, learn=True)
        assert r0 % TILE == 0 and chunk.is_contiguous() and chunk.shape[1] == self.m
        if self.ms == 0:
            return
        call('oriana_pack_count', ptr(chunk) + self.gd * chunk.element_size(), _XDTYPE[chunk.dtype], chunk.shape[0],
             self.ms, chunk.stride(0), r0 // TILE, self.ncb, ptr(self.tile_nnz), ptr(self.tile_rslots),
             ptr(self.tile_cslots), ptr(self.rslice), ptr(self.cslice), stream_ptr())

    def finish_count(self):
        nt = self.nrb * self.ncb
        i64 = dict(dtype=torch.int64, device=self.device)
        self.roff = torch.zeros(nt + 1, **i64)
        self.coff = torch.zeros(nt + 1, **i64)
        if nt:
            self.roff[1:] = torch.cumsum(self.tile_rslots[:nt].to(torch.int64), dim=0)
            self.coff[1:] = torch.cumsum(self.tile_cslots[:nt].to(torch.int64), dim=0)
        tot = torch.stack([self.roff[-1], self.coff[-1], self.tile_nnz[:max(nt, 1)].to(torch.int64).sum()]).tolist()
        self.rslots, self.cslots, self.nnz_sparse = int(tot[0]), int(tot[1]), int(tot[2]) if nt else 0
        self.nnz = self.nnz_sparse
        # padding slots are recognised by x == 0 / read row index 0: zero-fill before the fill pass
        self.rowrec = torch.zeros(max(self.rslots, 1), **i64)                                   # 8-byte records
        self.ridx = torch.zeros(max(self.cslots, 1), dtype=torch.uint8, device=self.device)

    def fill_chunk(self, chunk, r0, side=None, side_nz=None):
        chunk = self._permute(chunk, r0)
        if side is not None:
            side = self._permute(side, r0)
        assert r0 % TILE == 0 and chunk.is_contiguous()
        if self.dense is not None:
            if side is not None:
                raise _lib.OrianaHipError('a hybrid (dense-gene) layout carries no per-entry side matrix')
            self.dense.pack_chunk(chunk, r0)
        if self.ms == 0:
            return
        call('oriana_pack_fill', ptr(chunk) + self.gd * chunk.element_size(), _XDTYPE[chunk.dtype], chunk.shape[0], self.ms,
             chunk.stride(0), r0 // TILE, self.ncb, ptr(self.roff), ptr(self.coff), ptr(self.rslice), ptr(self.cslice),
             ptr(self.rowrec), ptr(self.ridx), ptr(side), side.stride(0) if side is not None else 0, ptr(side_nz),
             stream_ptr())

    def _build_col_work(self, target_items=None, width=1):
        """Work list of the column pass: (column block, row-block range) items of about equal
        COST, launched band of rows by band of rows.  A column block is `width` adjacent column tiles
        (oriana_col_block_tiles(K): for K <= 116 the kernel serves two tiles with one image of the row block).
        Genes differ widely in density, so uniform bands
        would leave the chip waiting for the densest column block; and the items that run at the same time
        should stage the SAME factor rows (at 1M cells the row-side factor is 400 MB, read once per column
        block: ordering the items by row range keeps the band being worked on in L2 / Infinity Cache --
        column pass 26.4 -> 25.0 ms at C4 against a longest-first order).  Cost of a tile = its longest column slice (the
        workgroup advances at the pace of its slowest wave) plus a fixed charge for staging the 256
        factor rows (measured on MI355X: ~1.45 us per slice iteration, ~3.2 us per tile).  The cut itself is
        oriana_plan_col_work (csrc/resident.hip: host code of the C ABI, for the compute units of the device at hand)."""
        nt = self.nrb * self.ncb
        if nt == 0 or self.cslots == 0:
            return None
        cs = self.cslice[:nt * 17].view(nt, 17).to(torch.int64)
        nit = ((cs[:, 1:] - cs[:, :-1]) // 64).max(dim=1).values                     # longest slice per tile
        nit = np.ascontiguousarray(nit.cpu().numpy().astype(np.int32))
        lib = _lib.load()
        cap = int(lib.oriana_plan_col_work_capacity(self.nrb, self.ncb, int(width)))
        items = np.empty((max(cap, 1), 3), dtype=np.int32)
        n_items = ctypes.c_int64(0)
        # (the pair's price: the maximum of the two slices -- with the constants fitted to it the sum measured 2.6 % slower on
        #  the column pass at C4, DESIGN_HISTORY.md 10 j; re-cut to whole rounds of the chip, DESIGN_HISTORY.md 10 l)
        call('oriana_plan_col_work', nit.ctypes.data, self.nrb, self.ncb, int(width), int(lib.oriana_device_cus()),
             int(target_items or 0), 1, 0, items.ctypes.data, cap, ctypes.addressof(n_items))
        return torch.from_numpy(items[:int(n_items.value)].copy()).to(self.device).contiguous()

    def col_work_for(self, K):
        """The work list matching the column tiles per work-group of the kernel that serves this K (cached)."""
        return self.col_work_width(int(_lib.load().oriana_col_block_tiles(int(K))) or 1)

    def col_work_width(self, width):
        if width not in self._col_work:
            self._col_work[width] = self._build_col_work(width=width)
        return self._col_work[width]

    def finish(self):
        self._col_work = {}
        # relative cost of the gene tiles on the row side (mean iterations of a 16-row slice + the staging of the tile's factor
        # rows, in iterations): where the row pass cuts gene ranges (oriana_row_pass_plan)
        self.gene_tile_cost = None
        if self.tile_rslots is not None and self.nrb * self.ncb > 0:
            per = self.tile_rslots[:self.nrb * self.ncb].view(self.nrb, self.ncb).to(torch.float64).mean(dim=0) / (16 * 64)
            stage = 2.0                      # (staging a tile's 256 factor rows, in slice iterations: flat between 0.5 and 4, DESIGN_HISTORY.md 10 l)
            self.gene_tile_cost = np.ascontiguousarray((per + stage).cpu().numpy(), dtype=np.float64)
        self.tile_rslots = self.tile_cslots = None
        if self.gd and self.col_perm is None:
            raise _lib.OrianaHipError('a hybrid layout needs an explicit gene order')
        # the sliced layout of a hybrid matrix is the sub-matrix of the packed genes [gd, m): its gene order is the
        # full one advanced by gd entries (so are the FV / C pointers the passes get, see zq_gap)
        self._struct = OrianaCounts(self.n, self.ms, self.nrb, self.ncb, self.nnz_sparse, self.rslots, self.cslots,
                                    ptr(self.roff), ptr(self.coff), ptr(self.rslice), ptr(self.cslice),
                                    ptr(self.rowrec), ptr(self.ridx),
                                    (ptr(self.col_perm) + 4 * self.gd) if self.col_perm is not None else None,
                                    ptr(self.row_perm))
        self.nnz = self.nnz_sparse + (self.dense.nnz if self.dense is not None else 0)
        return self

    @staticmethod
    def _gene_stats(chunks, m, device, reduce_fn, dense_density, n_total, min_share=0.0):
        """Per-gene non-zero counts over all row shards and, for a hybrid layout, the gene order and gd."""
        cn = torch.zeros(m, dtype=torch.int64, device=device)
        bad = torch.zeros(m, dtype=torch.int64, device=device) if dense_density else None
        n_local = 0
        for c in chunks():
            cn += (c != 0).sum(0)
            n_local += c.shape[0]
            if bad is not None:
                cf = c if c.dtype.is_floating_point else None
                b = (c < 0) | (c >= 65535)
                if cf is not None:
                    b |= cf != torch.floor(cf)
                bad += b.sum(0)
        if reduce_fn is not None:
            reduce_fn(cn)
            if bad is not None:
                reduce_fn(bad)
        if not dense_density:
            return cn, None, 0
        if n_total is None:
            nt = torch.tensor([n_local], dtype=torch.int64, device=device)
            if reduce_fn is not None:
                reduce_fn(nt)
            n_total = int(nt.item())
        order, gd = CountTiles.dense_order(cn, n_total, bad, dense_density, min_share)
        return cn, order, gd

    @classmethod
    def from_dense(cls, X, device='cuda', chunk_bytes=1 << 30, side=None, sort_cols=True, reduce_fn=None,
                   sort_rows=False, dense_density=None, n_total=None, dense_min_share=0.0):
        """Pack a dense (n, m) matrix (NumPy or torch, host or device).  `side`: optional dense
        (n, m) float32 DEVICE matrix gathered at the non-zeros (returned as .side_nz, row-side
        slots).  `reduce_fn`: sums the per-gene counts over row shards (all-reduce) so that every
        rank packs the genes in the same order.  `dense_density`: build a HYBRID layout -- genes expressed in at
        least this share of the cells (of all shards: `n_total`) go to a dense block evaluated on the matrix cores
        (csrc/dense_pass.hip; pCMF only, K with oriana_dense_supported)."""
        n, m = X.shape
        dev = torch.device(device)
        if n == 0 or m == 0:
            self = cls(n, m, device)
            self.finish_count()
            return self.finish()
        rows = max(TILE, (chunk_bytes // max(1, m * 8)) // TILE * TILE)
        gd, order = 0, None
        if sort_cols or dense_density:
            cn, order, gd = cls._gene_stats(lambda: (_as_device_chunk(X, r0, min(n, r0 + rows), dev) for r0 in range(0, n, rows)),
                                            m, dev, reduce_fn, dense_density if side is None else None, n_total, dense_min_share)
        self = cls(n, m, device, gd=gd)
        self.dense_density = dense_density if gd else None
        self.sort_rows = bool(sort_rows)
        if order is not None:
            self.col_perm = order.to(torch.int32).contiguous()
        elif sort_cols:
            self.set_col_order(cn)
        for r0 in range(0, n, rows):
            self.count_chunk(_as_device_chunk(X, r0, min(n, r0 + rows), self.device), r0)
        self.finish_count()
        if side is not None:
            self.side_nz = torch.zeros(max(self.rslots, 1), dtype=torch.float32, device=self.device)
        for r0 in range(0, n, rows):
            r1 = min(n, r0 + rows)
            self.fill_chunk(_as_device_chunk(X, r0, r1, self.device), r0,
                            side[r0:r1] if side is not None else None, self.side_nz)
        return self.finish()

    @classmethod
    def from_chunks(cls, n, m, chunk_fn, chunk_rows, device='cuda', sort_cols=True, reduce_fn=None, sort_rows=False,
                    dense_density=None, n_total=None, dense_min_share=0.0):
        """Passes over `chunk_fn(r0, r1) -> dense device tensor` (deterministic generator):
        per-gene counts (when sort_cols), tile counts, fill.  `dense_density`, `n_total`: as from_dense."""
        assert chunk_rows % TILE == 0
        gd, order = 0, None
        if (sort_cols or dense_density) and n > 0 and m > 0:
            cn, order, gd = cls._gene_stats(lambda: (chunk_fn(r0, min(n, r0 + chunk_rows)) for r0 in range(0, n, chunk_rows)),
                                            m, torch.device(device), reduce_fn, dense_density, n_total, dense_min_share)
        self = cls(n, m, device, gd=gd)
        self.dense_density = dense_density if gd else None
        self.sort_rows = bool(sort_rows)
        if order is not None:
            self.col_perm = order.to(torch.int32).contiguous()
        elif sort_cols and n > 0 and m > 0:
            self.set_col_order(cn)
        for r0 in range(0, n, chunk_rows):
            self.count_chunk(chunk_fn(r0, min(n, r0 + chunk_rows)).contiguous(), r0)
        self.finish_count()
        for r0 in range(0, n, chunk_rows):
            self.fill_chunk(chunk_fn(r0, min(n, r0 + chunk_rows)).contiguous(), r0)
        return self.finish()

    @classmethod
    def from_scipy(cls, A, device='cuda', chunk_rows=8192, sort_cols=True, reduce_fn=None, dense_density=None, n_total=None,
                   dense_min_share=0.0):
        """Pack a SciPy sparse (n, m) count matrix: row chunks of the CSR form are expanded on the
        device (chunk_rows x m floats at a time), so neither host nor device ever holds the dense
        matrix (real single-cell matrices are > 90 % zeros; reference cmatrix.py:39-53 only offers
        the dense route)."""
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        A.sum_duplicates()
        n, m = A.shape
        dev = torch.device(device)
        indptr = A.indptr.astype(np.int64)

        def chunk_fn(r0, r1):
            lo, hi = int(indptr[r0]), int(indptr[r1])
            out = torch.zeros(r1 - r0, m, dtype=torch.float32, device=dev)
            if hi > lo:
                counts = torch.from_numpy(np.diff(indptr[r0:r1 + 1])).to(dev)
                rows = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), counts)
                cols = torch.from_numpy(A.indices[lo:hi].astype(np.int64)).to(dev)
                vals = torch.from_numpy(np.asarray(A.data[lo:hi], dtype=np.float32)).to(dev)
                out[rows, cols] = vals
            return out
        if n == 0 or m == 0:
            self = cls(n, m, dev)
            self.finish_count()
            return self.finish()
        return cls.from_chunks(n, m, chunk_fn, max(TILE, chunk_rows // TILE * TILE), dev, sort_cols=sort_cols,
                               reduce_fn=reduce_fn, dense_density=dense_density, n_total=n_total, dense_min_share=dense_min_share)

    @property
    def c_struct(self):
        """struct oriana_counts of a PURE sliced layout (every consumer that knows nothing of dense genes)."""
        if self.gd:
            raise _lib.OrianaHipError('this count matrix has a hybrid layout (dense genes on the matrix cores): only the pCMF '
                                      'responsibility pass and the count metrics read it; pack without dense_density for this use')
        return ctypes.byref(self._struct)

    @property
    def sparse_struct(self):
        """The sliced layout: the whole matrix, or the packed genes [gd, m) of a hybrid layout."""
        return ctypes.byref(self._struct)

    def bytes_resident(self):
        b = self.rslots * 8 + self.cslots + (self.nrb * self.ncb) * (2 * 17 * 4 + 2 * 8 + 4)
        return b + (self.dense.x.numel() * 2 if self.dense is not None else 0)

    def slot_efficiency(self):
        """(nnz / row-side slots, nnz / column-side slots) of the sliced layout: the share of lanes that carry a real entry."""
        return (self.nnz_sparse / max(self.rslots, 1), self.nnz_sparse / max(self.cslots, 1))

    # ---- debugging / tests ------------------------------------------------------------------
    _REC = np.dtype([('x', '<f4'), ('cdst', '<u2'), ('col', 'u1'), ('pad', 'u1')])

    def host_arrays(self):
        nt = self.nrb * self.ncb
        return dict(roff=self.roff.cpu().numpy(), coff=self.coff.cpu().numpy(),
                    rslice=self.rslice.cpu().numpy().view(np.uint32).reshape(-1, 17)[:max(nt, 1)],
                    cslice=self.cslice.cpu().numpy().view(np.uint32).reshape(-1, 17)[:max(nt, 1)],
                    rec=self.rowrec.cpu().numpy().view(self._REC), ridx=self.ridx.cpu().numpy())

    def to_dense(self):
        """Rebuild the dense float32 matrix on the host (tests only)."""
        X = np.zeros((self.nrb * TILE, self.ncb * TILE), dtype=np.float32)
        if self.nnz_sparse:
            h = self.host_arrays()
            for rb in range(self.nrb):
                for cb in range(self.ncb):
                    t = rb * self.ncb + cb
                    for sl in range(16):
                        a, b = int(h['rslice'][t, sl]), int(h['rslice'][t, sl + 1])
                        if b == a:
                            continue
                        seg = h['rec'][h['roff'][t] + a:h['roff'][t] + b]
                        slot = np.arange(b - a)
                        rows = rb * TILE + sl * 16 + ((slot & 63) >> 2)
                        keep = seg['x'] != 0
                        X[rows[keep], cb * TILE + seg['col'][keep].astype(np.int64)] = seg['x'][keep]
        X = X[:self.n, :self.ms]
        if self.dense is not None:
            X = np.concatenate([self.dense.to_dense(), X], axis=1)
        if self.row_perm is not None:
            out = np.zeros_like(X)
            out[self.row_perm.cpu().numpy()] = X
            X = out
        if self.col_perm is not None:
            out = np.zeros_like(X)
            out[:, self.col_perm.cpu().numpy()] = X
            X = out
        return X


class DenseBlock:
    """The densest genes of a hybrid layout (struct oriana_dense): uint16 counts in 32 x 32 blocks, in the register
    order of the matrix-core row kernel (csrc/dense_pass.hip)."""

    def __init__(self, n, gd, device):
        self.n, self.gd = int(n), int(gd)
        self.ngt = self.gd // 32
        self.nct = (self.n + TILE - 1) // TILE * 8          # allocated cell tiles (whole 256-row blocks)
        self.x = torch.zeros(self.nct * self.ngt * 1024, dtype=torch.uint16, device=device)
        self.nnz = 0
        self._struct = OrianaDense(self.n, self.gd, self.nct, ptr(self.x))

    def pack_chunk(self, chunk, r0):
        assert r0 % 32 == 0
        self.nnz += int(torch.count_nonzero(chunk[:, :self.gd]).item())
        call('oriana_dense_pack', ptr(chunk), _XDTYPE[chunk.dtype], chunk.shape[0], self.gd, chunk.stride(0), r0 // 32,
             ptr(self.x), stream_ptr())

    @property
    def c_struct(self):
        return ctypes.byref(self._struct)

    def to_dense(self):
        """(n, gd) float32 on the host, packed gene order (tests only)."""
        x = self.x.cpu().numpy().reshape(self.nct, self.ngt, 2, 64, 8)          # [ct][gt][v / 8][lane][v % 8]
        out = np.zeros((self.nct * 32, self.gd), dtype=np.float32)
        for v in range(16):
            for h in range(2):
                g = 8 * (v >> 2) + 4 * h + (v & 3)
                blk = x[:, :, v >> 3, 32 * h:32 * h + 32, v & 7]                 # [ct][gt][cell]
                out.reshape(self.nct, 32, self.ngt, 32)[:, :, :, g] = blk.transpose(0, 2, 1)
        return out[:self.n]


def dense_supported(K):
    return bool(_lib.load().oriana_dense_supported(int(K)))


# Default density threshold of the hybrid layout (FactorModel(dense_density='auto')): measured break-even of the
# matrix-core evaluation against the sliced layout on the benchmark's density profile (DESIGN_HISTORY.md section 10).
DENSE_DENSITY_DEFAULT = 0.2


def dense_density_default():
    e = os.environ.get('ORIANA_DENSE_DENSITY')
    if e is None or e == '' or e == 'auto':
        return DENSE_DENSITY_DEFAULT
    if e.lower() in ('0', 'off', 'none', 'no'):
        return None
    return float(e)


def auto_dense_density(n_total, m, K):
    """The threshold 'auto' stands for: the default above for a K the dense kernels are compiled for and a matrix of at
    least 2e8 entries (below that a sweep is a few dozen launches of microseconds of work each, and the five extra
    launches of the hybrid layout cost more than its kernels save), None otherwise."""
    dd = dense_density_default()
    if not dd or not dense_supported(K) or float(n_total) * float(m) < 2e8:
        return None
    return float(dd)


class ZWorkspace:
    """Scratch buffers of the responsibility pass for one (CountTiles, K)."""

    def __init__(self, ct, K, need_sw=False, need_srow=False):
        self.ct, self.K, self.Kp = ct, int(K), kpad(K)
        dev = ct.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.FU = torch.zeros(max(ct.n, 1), self.Kp, **f32)
        self.FV = torch.zeros(max(ct.m, 1), self.Kp, **f32)
        self.R = None          # allocated below: (n, Kp), or one slab per gene split of the row pass
        self.C = torch.zeros(max(ct.m, 1), self.Kp, **f32)
        # per-sweep scalars s_ij in column-side slots: padding slots must stay 0, hence zeros()
        self.s_cs = torch.zeros(max(ct.cslots, 1), **f32)
        self.sw_cs = torch.zeros(max(ct.cslots, 1), **f32) if need_sw else None
        self.s_rs = torch.zeros(max(ct.rslots, 1), **f32) if need_srow else None
        self.tile_flag = torch.zeros(max(ct.nrb * ct.ncb, 1), dtype=torch.int32, device=dev)
        # statistics of the row maxima of E[log U], E[log V] and the partial sums they are built from (zeroed ONCE)
        self.stats = torch.zeros(int(_lib.load().oriana_prep_scratch_bytes()) // 4, **f32)
        self.center_ptr = ptr(self.stats) + int(_lib.load().oriana_prep_center_offset())    # {sum, count} of E[log U] per factor (log sums)
        self.den_min_ptr = ptr(self.stats) + int(_lib.load().oriana_prep_den_threshold_offset())   # the row kernels' den threshold (float32, device)
        self.timer = None      # set to a KernelTimer to time the launches of a sweep
        # [r5] the cell side of the NEXT sweep's factor preparation, written by the Gamma update that produces E[log U]
        # (oriana_gamma_update_prep): FU in a second buffer (the gene-side kernels of the running sweep still read the
        # current one), the row maxima, the per-group partial statistics.  fu_pending: they describe the E[log U] at hand.
        self.prep_blocks = int(_lib.load().oriana_gamma_update_prep_blocks(max(ct.n, 0), int(K))) if ct.n > 0 else 0
        # (not for short matrices: below 2^20 elements the sweep is launch-bound and the one-element-per-lane Gamma kernel
        #  is the faster one, csrc/updates.hip gu_small; ORIANA_FUSED_PREP=off: A/B runs)
        if os.environ.get('ORIANA_FUSED_PREP', 'on') == 'off' or (os.environ.get('ORIANA_FUSED_PREP') != 'force' and ct.n * int(K) < (1 << 20)):
            self.prep_blocks = 0
        self.FU_alt = self.mu_u = self.upart = None
        self.fu_pending = False
        self.fu_source = 0      # data pointer of the E[log U] matrix the pending preparation was made from
        self._extra = {}
        self._clear_cache = {}
        # gene-range split of the row pass (struct oriana_row_split: short matrices split every row block, long ones the row
        # blocks of the last, partly filled round of the chip); row_gene_splits = slabs of R, row_slab_row0 = first row with
        # more than slab 0
        self.row_split = _lib.OrianaRowSplit()
        self.row_split.nfull, self.row_split.parts = max(ct.nrb, 0), 1
        self.row_split.edge[0], self.row_split.edge[1] = 0, max(ct.ncb, 0)
        if ct.ms > 0 and ct.n > 0 and os.environ.get('ORIANA_ROW_SPLIT', 'auto') != 'off':
            cost = getattr(ct, 'gene_tile_cost', None)
            call('oriana_row_pass_plan', ct.sparse_struct, int(K), cost.ctypes.data if cost is not None else None,
                 ctypes.byref(self.row_split))
        self.row_gene_splits = int(self.row_split.parts)
        self.row_slab_row0 = int(self.row_split.nfull) * TILE
        gs = self.row_gene_splits
        self.R = self._alloc_R(gs, self.row_slab_row0)
        if ct.dense is not None:
            if not dense_supported(K):
                raise _lib.OrianaHipError('K=%d has no dense-gene kernels (oriana_dense_supported): pack without dense_density' % K)
            d = ct.dense
            lib = _lib.load()
            self.dn_S = torch.zeros(d.nct * d.ngt * 1024, **f32)
            self.dn_flag = torch.zeros(max(d.nct * d.ngt, 1), dtype=torch.int32, device=dev)
            pv, pu = int(lib.oriana_dense_image_pieces(K, 0)), int(lib.oriana_dense_image_pieces(K, 1))
            self.dn_imgV = torch.empty(d.ngt * pv * 4, **f32)
            self.dn_imgU = torch.empty(max((ct.n + 31) // 32, 1) * pu * 4, **f32)
            gsp, csp = ctypes.c_int64(1), ctypes.c_int64(1)
            call('oriana_plan_dense_splits', ct.n, d.gd, int(lib.oriana_device_cus()), ctypes.addressof(gsp), ctypes.addressof(csp))
            self.dn_gene_splits, self.dn_cell_splits = int(gsp.value), int(csp.value)

    def _alloc_R(self, parts, row0):
        """The row sums: slab 0 for every row; slabs 1 .. parts - 1 for the rows from `row0` on only (the row blocks of the
        last round that the plan splits; a whole-grid split has row0 = 0) -- the kernels address slab p at a stride of
        n - row0 rows, so the buffer is (n + (parts - 1)(n - row0)) rows (0.8 GB less at C4 than `parts` full slabs)."""
        n = max(self.ct.n, 1)
        rows = n + (parts - 1) * max(self.ct.n - int(row0), 0)
        return torch.zeros(rows, self.Kp, dtype=torch.float32, device=self.ct.device)

    def dense_tail(self, nslab):
        """(first split 256-cell block, parts) of the dense row kernel: the split of the sliced row pass's last round when the
        row sums of this call have that many slabs (and the dense kernel no split of its own), else no split."""
        sp = self.row_split
        if sp.nfull > 0 and 1 < sp.parts == nslab and self.dn_gene_splits == 1 and sp.parts <= self.ct.dense.ngt:
            return int(sp.nfull), int(sp.parts)
        return 0, 1

    def set_row_split(self, nfull, parts, edges=None):
        """Replace the plan of oriana_row_pass_plan (tests, tuning runs): row blocks [0, nfull) whole, the others in `parts`
        gene ranges cut at `edges` (parts + 1 gene-tile indices; None = evenly)."""
        ct = self.ct
        sp = self.row_split
        sp.nfull, sp.parts = int(nfull), int(parts)
        if edges is None:
            sp.edge[0] = -1
        else:
            assert len(edges) == parts + 1 <= 9
            for i, e in enumerate(edges):
                sp.edge[i] = int(e)
        self.row_gene_splits = int(parts)
        self.row_slab_row0 = int(nfull) * TILE
        self.R = self._alloc_R(int(parts), self.row_slab_row0)

    def prep_outputs(self, packed_rows):
        """(FU_next, mu, upart) for oriana_gamma_update[_finalize]_prep, or None when the cell side's Gamma update cannot
        prepare the next sweep's factor: no vector kernel for this K, or -- `packed_rows` False: the update walks the
        rows in the caller's order -- a count layout with a row permutation."""
        if not self.prep_blocks or (not packed_rows and self.ct.row_perm is not None):
            return None
        if self.FU_alt is None:
            f32 = dict(dtype=torch.float32, device=self.ct.device)
            self.FU_alt = torch.zeros(max(self.ct.n, 1), self.Kp, **f32)
            self.mu_u = torch.zeros(max(self.ct.n, 1), **f32)
            self.upart = torch.zeros(4 * self.prep_blocks, **f32)
        return self.FU_alt, self.mu_u, self.upart

    def extra(self, name, rows):
        """Lazily allocated padded (rows, Kp) scratch factor / accumulator matrices."""
        t = self._extra.get(name)
        if t is None:
            t = torch.zeros(max(rows, 1), self.Kp, dtype=torch.float32, device=self.ct.device)
            self._extra[name] = t
        return t


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are launched on (torch's
    current stream, which is the stream handed to the C ABI)."""

    def __init__(self, prealloc=0):
        """`prealloc`: event pairs created up front (creating an event costs microseconds of host time: on a small
        matrix that is comparable with the launches being timed)."""
        self.records = []
        self._pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(int(prealloc))]

    def span(self, name):
        return _Span(self, name)

    def _events(self):
        if self._pool:
            return self._pool.pop()
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

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
        self.a, self.b = self.timer._events()
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.timer.records.append((self.name, self.a, self.b))


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def nzmask_tiles(mask, n, m):
    """The per-lane non-zero flags of csrc/dense_zi.hip (oriana_nzmask_tiles) from the oriana_nzmask_f32 words of an (n, m)
    matrix: built once per count matrix, handed to dropout_sweep."""
    words = int(_lib.load().oriana_nzmask_tiles_words(n, m))
    tiles = torch.zeros(max(words, 4), dtype=torch.int32, device=mask.device)
    call('oriana_nzmask_tiles', ptr(tiles), ptr(mask), n, m, stream_ptr())
    return tiles


def dropout_sweep(D, U, V, pi, mask, tiles, colsum, V_next, DV_next, scratch, arithmetic, n, m, K):
    """oriana_dropout_sweep_fused_tiles (zigap.py:130-136 + the next sweep's zigap.py:116) on device tensors; None -> NULL."""
    call('oriana_dropout_sweep_fused_tiles', ptr(D), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(tiles), ptr(colsum), ptr(V_next),
         ptr(DV_next), ptr(scratch), int(arithmetic), n, m, K, stream_ptr())


def _span(ws, name):
    t = getattr(ws, 'timer', None)
    return t.span(name) if t is not None else _NoSpan()


def factor_prep(F, logF, mask=None, mu=None, row_index=None):
    r, K = logF.shape
    assert logF.dtype == torch.float32 and logF.is_contiguous()
    call('oriana_factor_prep', ptr(F), ptr(mu), ptr(logF), ptr(mask), ptr(row_index), r, K, stream_ptr())
    return F


DETERMINISTIC = os.environ.get('ORIANA_DETERMINISTIC') == '1'
_det_scratch = {}


def set_deterministic(flag=True):
    """Debug mode: the per-gene sums of the column pass are combined in a fixed order (per work
    item slabs + an ordered reduction) instead of float atomics, so two runs give bit-identical results; compare
    against the default path to see what the atomics' order is worth (~1e-7).  Also ORIANA_DETERMINISTIC=1."""
    global DETERMINISTIC
    DETERMINISTIC = bool(flag)


def _clear_list(ws, tensors):
    """struct oriana_clear_list over `tensors` (cached per set of buffers: a sweep hands in the same ones every time)."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    key = tuple((t.data_ptr(), t.numel() * t.element_size()) for t in tensors)
    hit = ws._clear_cache.get(key)
    if hit is None:
        if len(key) > 8:
            raise ValueError('at most 8 buffers can ride on the factor preparation')
        cl = _lib.OrianaClearList()
        for i, (p, b) in enumerate(key):
            cl.ptr[i] = p
            cl.bytes[i] = b
        if len(ws._clear_cache) > 16:
            ws._clear_cache.clear()
        hit = ws._clear_cache[key] = cl
    return hit


def factor_prep_pair(ws, log_U_hat, log_V_hat, mask_v=None, clear=None):
    """FU, FV of the workspace from E[log U], E[log V] (+ S_tilde), validity test centred on the typical shifts of
    the two sides (oriana_factor_prep_pair): the sweeps drift along U c, V / c and only the sums matter.  `clear`: up to 8
    contiguous tensors zero-filled by the same launch (the outputs and scratch the passes accumulate into)."""
    ct = ws.ct
    cl = None
    if clear:
        for t in clear:
            if t is not None and not t.is_contiguous():
                raise ValueError('buffers on the clear list must be contiguous')
        cl = _clear_list(ws, clear)
    if ws.fu_pending and ws.fu_source != log_U_hat.data_ptr():
        ws.fu_pending = False               # (a caller's own E[log U]: the preparation at hand belongs to another matrix)
    if ws.fu_pending:
        # the cell side was prepared by the Gamma update that wrote this E[log U] (ZWorkspace.prep_outputs): the buffers swap,
        # this launch combines the statistics, prepares the gene side and overwrites the rejected cell rows
        ws.fu_pending = False
        ws.FU, ws.FU_alt = ws.FU_alt, ws.FU
        call('oriana_factor_prep_pair_fused', ptr(ws.FU), ptr(ws.mu_u), ptr(ws.upart), ws.prep_blocks, ptr(ws.FV), ptr(log_V_hat),
             ptr(mask_v), ptr(ct.col_perm), ct.n, ct.m, ws.K, ptr(ws.stats), ctypes.byref(cl) if cl is not None else None, stream_ptr())
        return
    if cl is not None:
        call('oriana_factor_prep_pair_clear', ptr(ws.FU), ptr(ws.FV), ptr(log_U_hat), ptr(log_V_hat), ptr(mask_v),
             ptr(ct.row_perm), ptr(ct.col_perm), ct.n, ct.m, ws.K, ptr(ws.stats), ctypes.byref(cl), stream_ptr())
        return
    call('oriana_factor_prep_pair', ptr(ws.FU), ptr(ws.FV), ptr(log_U_hat), ptr(log_V_hat), ptr(mask_v), ptr(ct.row_perm),
         ptr(ct.col_perm), ct.n, ct.m, ws.K, ptr(ws.stats), stream_ptr())


# analysis runs (tools/parity_report.py): float64 accumulators and one rounding for the per-gene sums -- '1': all of them, 'log': the
# centred log sums of the sparse models only (C2), 'zj': everything but those
_COL_F64 = os.environ.get('ORIANA_COL_F64', '')


def col_pass(ct, s_cs, G, C, K, C_ptr=None, what='zj'):
    """C += s G over the sliced layout.  `C_ptr`: device address of the first gene row of the sliced layout inside C
    (hybrid layouts: gd rows in).  `what`: 'zj' (the per-gene sums of gap.py:80) or 'log' (the centred log sums)."""
    w = ct.col_work_for(K)
    Cp = ptr(C) if C_ptr is None else C_ptr
    if _COL_F64 == '1' or _COL_F64 == what:
        call('oriana_col_pass_f64acc', ct.sparse_struct, ptr(s_cs), ptr(G), Cp, K, stream_ptr())
        return
    if DETERMINISTIC and w is not None:
        nbytes = int(_lib.load().oriana_col_pass_det_scratch_bytes(int(K), int(w.shape[0])))
        key = (ct.device, nbytes)
        if key not in _det_scratch:
            _det_scratch.clear()
            _det_scratch[key] = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=ct.device)
        call('oriana_col_pass_det', ct.sparse_struct, ptr(s_cs), ptr(G), Cp, K, ptr(w), w.shape[0], ptr(_det_scratch[key]),
             stream_ptr())
        return
    call('oriana_col_pass', ct.sparse_struct, ptr(s_cs), ptr(G), Cp, K, ptr(w), 0 if w is None else w.shape[0], stream_ptr())


def col_pass_dual(ct, s_cs, G1, G2, C1, C2, K, goff=0):
    """C1 += s G1 and C2 += s G2 from one walk over the column-side stream (oriana_col_pass_dual).  Returns False when
    the two factor images do not fit in LDS (or in the deterministic debug mode): the caller runs two column passes.
    `goff`: byte offset of the sliced part's first gene row inside C1 / C2 (hybrid layouts)."""
    if DETERMINISTIC or not _FUSE_SPARSE_COLS or _COL_F64:
        return False                        # (the analysis mode runs the two sums as two passes)
    w = ct.col_work_width(1)
    if w is None:
        return False
    rc = _lib.load().oriana_col_pass_dual(ct.sparse_struct, ptr(s_cs), ptr(G1), ptr(G2), ptr(C1) + goff, ptr(C2) + goff, K, ptr(w),
                                          w.shape[0], stream_ptr())
    if rc not in (0, -2):
        raise _lib.OrianaHipError('oriana_col_pass_dual failed with code %d' % rc)
    return rc == 0


def zq_gap(ws, Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, phase='all', finalize_rows=True, finalize_cols=True, clear=(),
           zj_packed=False, on_segment=None):
    """GaP.compute_Z_q_expectations (reference gap.py:67-80) on the resident tiles: outputs first,
    zero-filled by the callee, returns None.  `phase`: 'rows' stops once Z_hat_i is final (factor
    preparation, row pass, slow path, row-side finalize), 'cols' does the rest (column pass, gene-side finalize):
    the cell-side Gamma update only needs Z_hat_i, so a sharded sweep runs it between the two and has every
    partial of its single exchange ready when the column pass ends.
    finalize_rows / finalize_cols = False: the caller completes Z_hat_i / Z_hat_j itself (the pCMF sweep folds
    Z += F * R into its Gamma updates, oriana_gamma_update_finalize); until then they hold the slow path's additions
    only.  `clear`: further buffers (at most 3) zero-filled by the factor preparation's launch.
    zj_packed (the row-sharded pCMF sweep): Z_hat_j is kept in the PACKED gene order -- the slow paths index it by the packed
    gene, the gene-side finalize writes it without the permutation -- so that the dense genes [0, gd) and the sliced genes
    [gd, m) are two contiguous segments of the exchange buffer; `on_segment(lo, hi)` is called as soon as the rows
    [lo, hi) of Z_hat_j are final (the sliced segment after the sliced column pass, i.e. BEFORE the dense gene-side
    kernel runs: its all-reduce travels under that kernel)."""
    ct, K = ws.ct, ws.K
    st = stream_ptr()
    dn = ct.dense
    gd = ct.gd
    # hybrid layout: the sliced layout covers the packed genes [gd, m) -- its FV / C rows start gd rows in
    FVs, Cs = ptr(ws.FV) + 4 * gd * ws.Kp, ptr(ws.C) + 4 * gd * ws.Kp
    if phase in ('all', 'rows'):
        _check_f32(Z_hat_i, (ct.n, K)); _check_f32(Z_hat_j, (ct.m, K))
        _check_f32(log_U_hat, (ct.n, K)); _check_f32(log_V_hat, (ct.m, K))
        # short matrices: the gene tiles of a row block are split over several work-groups, which add into R
        gs = ws.row_gene_splits
        zero_R = ws.R if ct.ms == 0 else None              # (no sliced part: the dense row pass adds into it)
        factor_prep_pair(ws, log_U_hat, log_V_hat, clear=(Z_hat_i, Z_hat_j, ws.C, ws.tile_flag, zero_R) + tuple(clear))
        if ct.ms > 0:
            with _span(ws, 'row_pass'):
                call('oriana_row_pass_general', ct.sparse_struct, ptr(ws.FU), FVs, None, None, ptr(ws.R), ptr(ws.s_cs), None, None,
                     ptr(ws.tile_flag), K, ctypes.byref(ws.row_split), ws.den_min_ptr, st)
        if dn is not None:
            with _span(ws, 'dense_images'):
                call('oriana_dense_images', ptr(ws.dn_imgV), ptr(ws.FV), gd, K, 0, st)
            with _span(ws, 'dense_row'):
                # (the blocks of the chip's last round split as the sliced row pass split them: same rows, same slabs of R)
                tail = ws.dense_tail(gs)
                call('oriana_dense_row_pass_tail', dn.c_struct, ptr(ws.FU), ptr(ws.dn_imgV), ptr(ws.R), ptr(ws.dn_S),
                     ptr(ws.dn_flag), K, ws.dn_gene_splits, tail[0], tail[1], ws.den_min_ptr, st)
        with _span(ws, 'fixup'):
            if ct.ms > 0:
                # (packed Z_hat_j: the sliced part's packed gene 0 is row gd of the buffer)
                call('oriana_fixup', ct.sparse_struct, ptr(ws.tile_flag), ptr(ws.s_cs), None, None, ptr(log_U_hat),
                     ptr(log_V_hat), None, None, None, None, ptr(Z_hat_i), ptr(Z_hat_j) + (4 * gd * K if zj_packed else 0), None, K,
                     8 if zj_packed else 0, st)
            if dn is not None:
                call('oriana_dense_fixup_variant', dn.c_struct, ptr(ws.dn_flag), ptr(ws.dn_S), ptr(log_U_hat), ptr(log_V_hat),
                     ptr(ct.row_perm), ptr(ct.col_perm), ptr(Z_hat_i), ptr(Z_hat_j), None, None, None, None, K,
                     1 if zj_packed else 0, st)
        if finalize_rows:
            call('oriana_finalize_slabs_from', ptr(Z_hat_i), ptr(ws.FU), ptr(ws.R), gs, ws.row_slab_row0, ptr(ct.row_perm), ct.n, K, st)
    if phase in ('all', 'cols'):
        if ct.ms > 0:
            with _span(ws, 'col_pass'):
                col_pass(ct, ws.s_cs, ws.FU, ws.C, K, C_ptr=Cs)
            if zj_packed:                   # the sliced genes' rows of Z_hat_j are final: hand them to the exchange
                call('oriana_finalize', ptr(Z_hat_j) + 4 * gd * K, FVs, Cs, None, None, ct.ms, K, 1, st)
                if on_segment is not None:
                    on_segment(gd, ct.m)
        if dn is not None:
            with _span(ws, 'dense_images'):
                call('oriana_dense_images', ptr(ws.dn_imgU), ptr(ws.FU), ct.n, K, 1, st)
            with _span(ws, 'dense_col'):
                call('oriana_dense_col_pass', dn.c_struct, ptr(ws.dn_imgU), ptr(ws.dn_S), ptr(ws.C), K,
                     ws.dn_cell_splits, st)
            if zj_packed:
                call('oriana_finalize', ptr(Z_hat_j), ptr(ws.FV), ptr(ws.C), None, None, gd, K, 1, st)
                if on_segment is not None:
                    on_segment(0, gd)
        if finalize_cols and not zj_packed:
            call('oriana_finalize', ptr(Z_hat_j), ptr(ws.FV), ptr(ws.C), None, ptr(ct.col_perm), ct.m, K, 1, st)


def zq_gap_stateless(Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X):
    """oriana_zq_gap_f32: the reference signature on dense device tensors (packs X per call)."""
    for t in (Z_hat_i, Z_hat_j, log_U_hat, log_V_hat, X):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() != 2 or not t.is_contiguous():
            raise TypeError('expected 2-D C-contiguous float32 device tensors')
    n, K = log_U_hat.shape
    m = log_V_hat.shape[0]
    _check_f32(Z_hat_i, (n, K)); _check_f32(Z_hat_j, (m, K)); _check_f32(log_V_hat, (m, K)); _check_f32(X, (n, m))
    ws, base, nbytes = _stateless_ws(n, m, K, X)
    call('oriana_zq_gap_f32', ptr(Z_hat_i), ptr(Z_hat_j), ptr(log_U_hat), ptr(log_V_hat), ptr(X), n, m, K,
         base, nbytes, stream_ptr())


# (module switches for tests: the two-kernel forms of the sparse row / column phase, which every Kp > 64 takes anyway)
_FUSE_SPARSE_ROWS = True
_FUSE_SPARSE_COLS = True


def zq(ws, Z_i, Z_j, Z_log, log_U_hat, log_V_hat, S_tilde=None, S_hat=None, dq=None, w_nz=None, phase='all'):
    """The four loop nests on the resident tiles.  `w_nz` = D_hat at the stored entries (row-side
    slots, CountTiles.side_nz); None means 1, which is always the case inside the models
    (zigap.py:135 sets the dropout posterior of every non-zero count to 1 in float32):
      Z_i[i,k]   = sum_j [S_hat[j,k]] r_ijk                    (gap.py:79, sparse_gap.py:95)
      Z_j[j,k]   = sum_i [dq[i,k]] r_ijk                       (gap.py:80; dq = D_hat[:, :K], zigap.py:94)
      Z_log[j,k] = sum_i r_ijk (lu_ik + lv_jk)                 (zigap.py:95) -- skipped when Z_log is None
    with r_ijk = x_ij e_k / sum_k e_k, e_k = exp(lu_ik + lv_jk) [S_tilde[j,k]].  Outputs first,
    zero-filled here, float32 device tensors.  `phase` as in zq_gap ('rows': everything Z_i needs; 'cols': the
    per-gene sums; both phases must get the same arguments).
    HYBRID layouts (every nest without per-entry weights): the sliced kernels cover the packed genes [gd, m) -- their
    gene-side pointers start gd rows in --, the dense-gene kernels the first gd: den against the (masked) FV image, the
    accumulation against FV * S_hat (oriana_dense_images2), one gene-side pass per per-gene sum (FU [* dq], and the
    centred E[log U]-weighted factor of the log sums)."""
    ct, K = ws.ct, ws.K
    n, m = ct.n, ct.m
    sparse = S_hat is not None
    st = stream_ptr()
    dn, gd = ct.dense, ct.gd
    if dn is not None and w_nz is not None:
        raise _lib.OrianaHipError('a hybrid (dense-gene) layout carries no per-entry weights: pack without dense_density for this use')
    cst = ct.sparse_struct                       # (== the whole layout when there are no dense genes)
    goff = 4 * gd * ws.Kp                        # byte offset of the sliced part's first gene row in FV, C, ...
    have_sliced = ct.ms > 0 or dn is None
    if phase in ('all', 'rows'):
        _check_f32(Z_i, (n, K)); _check_f32(Z_j, (m, K)); _check_f32(log_U_hat, (n, K)); _check_f32(log_V_hat, (m, K))
        if w_nz is not None and ws.sw_cs is None:
            ws.sw_cs = torch.zeros(max(ct.cslots, 1), dtype=torch.float32, device=ct.device)
    sw_cs = ws.sw_cs if w_nz is not None else None
    if phase in ('all', 'rows'):
        zero_R = ws.R if not have_sliced else None          # (no sliced part: the dense row pass adds into it)
        factor_prep_pair(ws, log_U_hat, log_V_hat, mask_v=S_tilde, clear=(Z_i, Z_j, ws.C, ws.tile_flag, Z_log, zero_R))
        # sparse models: the S_hat-weighted row sums (sparse_gap.py:95).  Where two factor images fit in LDS (Kp <= 64)
        # they come out of the row pass itself (dot product against FV, accumulation against FV * S_hat); otherwise the
        # pass leaves s in row-side slots and a second row product follows.
        fused = False
        F2 = None
        gs = ws.row_gene_splits                     # gene ranges per row block (slabs of R), see oriana_row_pass_gene_splits
        nslab = 1
        if sparse:
            F2 = ws.extra('FVS', m)
            call('oriana_scale_factor', ptr(F2), ptr(ws.FV), ptr(S_hat), ptr(ct.col_perm), m, K, 0, st)
        if have_sliced:
            if sparse and _FUSE_SPARSE_ROWS:
                with _span(ws, 'row_pass'):
                    rc = _lib.load().oriana_row_pass_general(cst, ptr(ws.FU), ptr(ws.FV) + goff, ptr(F2) + goff, ptr(w_nz), ptr(ws.R),
                                                             ptr(ws.s_cs), ptr(sw_cs), None, ptr(ws.tile_flag), K, ctypes.byref(ws.row_split), ws.den_min_ptr, st)
                if rc not in (0, -2):
                    raise _lib.OrianaHipError('oriana_row_pass_general failed with code %d' % rc)
                fused = rc == 0
            if not fused:
                if sparse and ws.s_rs is None:   # row-side copy of s for the second row product (lazy: the fused form never needs it)
                    ws.s_rs = torch.zeros(max(ct.rslots, 1), dtype=torch.float32, device=ct.device)
                with _span(ws, 'row_pass'):
                    call('oriana_row_pass_general', cst, ptr(ws.FU), ptr(ws.FV) + goff, None, ptr(w_nz), ptr(ws.R), ptr(ws.s_cs),
                         ptr(sw_cs), ptr(ws.s_rs) if sparse else None, ptr(ws.tile_flag), K, ctypes.byref(ws.row_split), ws.den_min_ptr, st)
            if fused or not sparse:
                nslab = gs                       # (the second row product of the unfused sparse form writes one slab)
            with _span(ws, 'fixup'):
                call('oriana_fixup', cst, ptr(ws.tile_flag), ptr(ws.s_cs), ptr(sw_cs),
                     ptr(ws.s_rs) if (sparse and not fused) else None,
                     ptr(log_U_hat), ptr(log_V_hat), ptr(S_tilde), ptr(S_hat), ptr(w_nz), ptr(dq), ptr(Z_i), ptr(Z_j), ptr(Z_log),
                     K, (1 if sparse else 0) | (2 if w_nz is not None else 0) | (4 if dq is not None else 0), st)
            if sparse and not fused:
                with _span(ws, 'row_spmm'):
                    call('oriana_row_spmm', cst, ptr(ws.s_rs), ptr(w_nz), ptr(F2) + goff, ptr(ws.R), K, st)
        if dn is not None:
            # (after the sliced kernels: the dense row kernel ADDS its sums into R)
            with _span(ws, 'dense_images'):
                call('oriana_dense_images2', ptr(ws.dn_imgV), ptr(ws.FV), ptr(F2), gd, K, 0, st)
            with _span(ws, 'dense_row'):
                tail = ws.dense_tail(nslab)
                call('oriana_dense_row_pass_tail', dn.c_struct, ptr(ws.FU), ptr(ws.dn_imgV), ptr(ws.R), ptr(ws.dn_S), ptr(ws.dn_flag), K,
                     ws.dn_gene_splits, tail[0], tail[1], ws.den_min_ptr, st)
            with _span(ws, 'fixup'):
                call('oriana_dense_fixup_variant', dn.c_struct, ptr(ws.dn_flag), ptr(ws.dn_S), ptr(log_U_hat), ptr(log_V_hat),
                     ptr(ct.row_perm), ptr(ct.col_perm), ptr(Z_i), ptr(Z_j), ptr(Z_log), ptr(dq), ptr(S_tilde), ptr(S_hat), K, 0, st)
        call('oriana_finalize_slabs_from', ptr(Z_i), ptr(ws.FU), ptr(ws.R), nslab, ws.row_slab_row0, ptr(ct.row_perm), n, K, st)
        if Z_log is not None:
            # E[log U]-weighted row factor of the log sums: built NOW, from the pre-update E[log U] (the caller
            # may run the cell-side update between the two phases)
            call('oriana_log_center', ws.center_ptr, ptr(ws.FU), ptr(log_U_hat), ptr(Z_i), ptr(ct.row_perm), n, K, st)
            call('oriana_scale_factor_centered', ptr(ws.extra('GL', n)), ptr(ws.FU), ptr(log_U_hat), ws.center_ptr, ptr(ct.row_perm), n, K, st)
    if phase == 'rows':
        return

    def dense_cols(Gmat, Cmat):
        """The dense genes' share of  Cmat += s Gmat  (gene-side kernel over the cell images of Gmat)."""
        with _span(ws, 'dense_images'):
            call('oriana_dense_images', ptr(ws.dn_imgU), ptr(Gmat), n, K, 1, st)
        with _span(ws, 'dense_col'):
            call('oriana_dense_col_pass', dn.c_struct, ptr(ws.dn_imgU), ptr(ws.dn_S), ptr(Cmat), K, ws.dn_cell_splits, st)
    # per-gene sums: weighted by D_hat[i, j] (sw), or -- zigap.py:94 -- by D_hat[i, k] on the plain s
    G, s_for_j = ws.FU, (sw_cs if sw_cs is not None else ws.s_cs)
    if dq is not None:
        G, s_for_j = ws.extra('GQ', n), ws.s_cs
        call('oriana_scale_factor', ptr(G), ptr(ws.FU), ptr(dq), ptr(ct.row_perm), n, K, 0, st)
    # the log sums (zigap.py:95) need  sum_i s FU  and  sum_i s FU E[log U]  over the same stream: where two factor images
    # fit in LDS both come from ONE column pass
    dual_done = False
    if Z_log is not None:
        _check_f32(Z_log, (m, K))
        s_log = sw_cs if sw_cs is not None else ws.s_cs
        G2 = ws.extra('GL', n)
        C2 = ws.extra('C2', m)
        C2.zero_()
        if dq is None and have_sliced:          # the per-gene sums and the log sums share factor and stream
            with _span(ws, 'col_pass'):
                dual_done = col_pass_dual(ct, s_log, ws.FU, G2, ws.C, C2, K, goff=goff)
    if not dual_done and have_sliced:
        with _span(ws, 'col_pass'):
            col_pass(ct, s_for_j, G, ws.C, K, C_ptr=ptr(ws.C) + goff)
    if dn is not None:
        dense_cols(G, ws.C)
    call('oriana_finalize', ptr(Z_j), ptr(ws.FV), ptr(ws.C), None, ptr(ct.col_perm), m, K, 1, st)
    if Z_log is not None:
        if dq is not None:                      # the log sums use the D_hat[i, j]-weighted column sums
            ws.C.zero_()
            if have_sliced:
                with _span(ws, 'col_pass_log'):
                    dual_done = col_pass_dual(ct, s_log, ws.FU, G2, ws.C, C2, K, goff=goff)
                    if not dual_done:
                        col_pass(ct, s_log, ws.FU, ws.C, K, C_ptr=ptr(ws.C) + goff)
            if dn is not None:
                dense_cols(ws.FU, ws.C)
        if not dual_done and have_sliced:
            with _span(ws, 'col_pass_log'):
                col_pass(ct, s_log, G2, C2, K, C_ptr=ptr(C2) + goff, what='log')
        if dn is not None:
            dense_cols(G2, C2)
        call('oriana_finalize_zlog', ptr(Z_log), ptr(ws.FV), ptr(C2), ptr(ws.C), ptr(log_V_hat), ws.center_ptr, ptr(ct.col_perm), m, K, st)


def _stateless_ws(n, m, K, X):
    kpad(K)
    nnz = int(torch.count_nonzero(X).item()) if X.numel() else 0
    nbytes = int(_lib.load().oriana_zq_workspace_bytes(n, m, K, nnz + 64))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=X.device)
    return ws, (ws.data_ptr() + 255) // 256 * 256, nbytes


def zq_dense(Z_i, Z_j, Z_log, log_U_hat, log_V_hat, X, S_tilde=None, S_hat=None, D_hat=None, quirk=False):
    """The reference's loop-nest signatures on dense float32 device tensors (zigap.py:79-95,
    sparse_gap.py:81-97, sparse_zigap.py:100-116) through the stateless C-ABI entries
    oriana_zq_zigap_f32 / oriana_zq_sparse_gap_f32 / oriana_zq_sparse_zigap_f32: X is packed (and
    D_hat gathered at the non-zeros) on every call, like the reference re-casts X on every call."""
    for t in (Z_i, Z_j, log_U_hat, log_V_hat, X) + tuple(a for a in (Z_log, S_tilde, S_hat, D_hat) if a is not None):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() != 2 or not t.is_contiguous():
            raise TypeError('expected 2-D C-contiguous float32 device tensors')
    n, K = log_U_hat.shape
    m = log_V_hat.shape[0]
    _check_f32(Z_i, (n, K)); _check_f32(Z_j, (m, K)); _check_f32(log_V_hat, (m, K)); _check_f32(X, (n, m))
    if Z_log is not None:
        _check_f32(Z_log, (m, K))
    for t in (S_tilde, S_hat):
        if t is not None:
            _check_f32(t, (m, K))
    if D_hat is not None:
        _check_f32(D_hat, (n, m))
    if quirk and (D_hat is None or K > m):
        raise ValueError('the zigap.py:94 quirk needs D_hat and K <= number of genes')
    ws, base, nbytes = _stateless_ws(n, m, K, X)
    st = stream_ptr()
    if S_hat is None and D_hat is None:
        if Z_log is not None:
            raise ValueError('the pCMF loop nest (gap.py:67-80) has no log-sum output')
        call('oriana_zq_gap_f32', ptr(Z_i), ptr(Z_j), ptr(log_U_hat), ptr(log_V_hat), ptr(X), n, m, K, base, nbytes, st)
    elif S_hat is None:
        call('oriana_zq_zigap_f32', ptr(Z_i), ptr(Z_j), ptr(Z_log), ptr(log_U_hat), ptr(log_V_hat), ptr(D_hat), ptr(X),
             n, m, K, 1 if quirk else 0, base, nbytes, st)
    elif D_hat is None:
        call('oriana_zq_sparse_gap_f32', ptr(Z_i), ptr(Z_j), ptr(Z_log), ptr(log_U_hat), ptr(log_V_hat), ptr(S_tilde),
             ptr(S_hat), ptr(X), n, m, K, base, nbytes, st)
    else:
        if quirk:
            raise ValueError('sparse_zigap.py:114-116 has no D_hat[i, k] quirk')
        call('oriana_zq_sparse_zigap_f32', ptr(Z_i), ptr(Z_j), ptr(Z_log), ptr(log_U_hat), ptr(log_V_hat), ptr(S_tilde),
             ptr(S_hat), ptr(D_hat), ptr(X), n, m, K, base, nbytes, st)
    del ws


def _check_f32(t, shape):
    """numba's explicit signature raises TypeError on dtype / ndim mismatch (gap.py:67)."""
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError('expected a 2-D float32 device tensor')
    if not t.is_contiguous():
        raise TypeError('expected a C-contiguous tensor')
    if tuple(t.shape) != tuple(shape):
        raise ValueError('shape mismatch: %s vs %s' % (tuple(t.shape), tuple(shape)))

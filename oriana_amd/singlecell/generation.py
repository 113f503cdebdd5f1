# -*- coding: utf-8 -*-
"""Synthetic count matrices, generated on the device one row chunk at a time.

Row-blocked restatement of the reference generator (oriana/singlecell/generation.py:8-86):
block-structured Gamma factors U (cells x k) and V (genes x k), per-gene expression probability
pi_d ~ Beta(1, 1/z - 1), dropout mask D ~ Bernoulli(pi_d) and X = floor(D * U V^T) (no Poisson
draw, generation.py:85-86).  Gamma(1, scale) is an exponential, so only uniform / exponential
device RNG is needed.  The distribution is the reference's; the random streams are not (the
reference draws from NumPy's global state on the host).

A chunk is a pure function of (seed, row range): the tiled packer calls it twice (count, fill).

``shuffle=True`` applies the step the reference leaves as a TODO (generation.py:75, "shuffle cells and shuffle
genes"): the block structure of U is assigned through a seeded permutation of the cells and the rows of V / pi_d
through a seeded permutation of the genes, so that neither the cell groups nor the structured genes are
contiguous.  ``labels()`` follows the permutation.
"""
import math

import torch

__all__ = ['SyntheticCounts']


def _block_edges(total, n_groups):
    step = total // n_groups
    return [g * step for g in range(n_groups)] + [total]          # generation.py:9-12


class SyntheticCounts:
    """X = floor(D * U V^T) for cells [row0, row0 + n) of an (n_total, m) problem."""

    def __init__(self, n_total, m, k, seed, device='cuda', zero_inflation_level=0.5, sparsity_degree_in_v=0.5,
                 beta=80.0, theta=0.8, n_groups=2, row0=0, n=None, shuffle=False):
        self.n_total, self.m, self.k = int(n_total), int(m), int(k)
        self.row0 = int(row0)
        self.n = int(n) if n is not None else self.n_total - self.row0
        self.seed = int(seed)
        self.device = torch.device(device)
        self.theta, self.n_groups = theta, n_groups
        self.shuffle = bool(shuffle)
        self.cell_perm = self.gene_perm = None
        if self.shuffle:                                          # replicated, like every quantity drawn here
            gp = torch.Generator(device=self.device)
            gp.manual_seed(self.seed * 7 + 3)
            self.cell_perm = torch.randperm(self.n_total, generator=gp, device=self.device)
            self.gene_perm = torch.randperm(self.m, generator=gp, device=self.device)
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed)                                  # replicated quantities: same on every rank
        # generate_u (generation.py:8-37): per-group scale alpha / k, alpha in {100, 250}
        choice = torch.randint(0, 2, (n_groups,), generator=g, device=self.device)
        self.alpha = torch.where(choice == 0, 100.0, 250.0).to(torch.float32) / self.k
        self.alpha_bar = float(self.alpha.mean().item())
        self.u_row_edges = _block_edges(self.n_total, n_groups)
        self.k_edges = _block_edges(self.k, n_groups)
        # generate_v (generation.py:40-65)
        m0 = int(round(self.m * sparsity_degree_in_v))
        v_row_edges = _block_edges(m0, n_groups)
        scale = torch.full((self.m, self.k), (1.0 - theta) * beta, dtype=torch.float32, device=self.device)
        for grp in range(n_groups):
            scale[v_row_edges[grp]:v_row_edges[grp + 1], self.k_edges[grp]:self.k_edges[grp + 1]] = beta
        if self.gene_perm is not None:
            scale = scale[self.gene_perm]                         # the structured genes are no longer the first m0
        self.V = torch.empty(self.m, self.k, dtype=torch.float32, device=self.device).exponential_(1.0, generator=g) * scale
        # pi_d ~ Beta(1, 1/z - 1) (generation.py:80): inverse CDF 1 - u^(1/b)
        b = 1.0 / zero_inflation_level - 1.0
        u = torch.rand(self.m, generator=g, device=self.device, dtype=torch.float64)
        self.pi_d = (1.0 - u.pow(1.0 / b)).to(torch.float32) if b > 0 else torch.ones(self.m, device=self.device)

    def _group_rows(self, r0, r1):
        """Position of local cells [r0, r1) in the block structure (their own index, or a permuted one)."""
        rows = torch.arange(self.row0 + r0, self.row0 + r1, device=self.device)
        return self.cell_perm[rows] if self.cell_perm is not None else rows

    def _u_scale(self, r0, r1):
        rows = self._group_rows(r0, r1)
        scale = torch.full((r1 - r0, self.k), (1.0 - self.theta) * self.alpha_bar, dtype=torch.float32, device=self.device)
        for grp in range(self.n_groups):
            inrow = (rows >= self.u_row_edges[grp]) & (rows < self.u_row_edges[grp + 1])
            blk = scale[:, self.k_edges[grp]:self.k_edges[grp + 1]]
            blk[inrow] = self.alpha[grp]
        return scale

    CELL = 4096     # rows per RNG cell: the data are a function of (seed, global row), not of the sharding

    def _cell_rng(self, cell, salt):
        g = torch.Generator(device=self.device)
        g.manual_seed((self.seed * 1000003 + cell * 7919 + salt) % (2 ** 62))
        return g

    def _cells(self, r0, r1):
        """Global RNG cells intersecting local rows [r0, r1): (cell, global row range inside it)."""
        g0, g1 = self.row0 + r0, self.row0 + r1
        for cell in range(g0 // self.CELL, (g1 + self.CELL - 1) // self.CELL):
            a = max(g0, cell * self.CELL)
            b = min(g1, (cell + 1) * self.CELL)
            yield cell, a, b

    def u_chunk(self, r0, r1):
        """The generating factor rows U[r0:r1] (generation.py:8-37), float32 (tests, experiments)."""
        out = torch.empty(r1 - r0, self.k, dtype=torch.float32, device=self.device)
        for cell, a, b in self._cells(r0, r1):
            g = self._cell_rng(cell, 17)
            c0 = cell * self.CELL
            U = torch.empty(self.CELL, self.k, dtype=torch.float32, device=self.device).exponential_(1.0, generator=g)
            out[a - self.row0 - r0:b - self.row0 - r0] = U[a - c0:b - c0] * self._u_scale(a - self.row0, b - self.row0)
        return out

    def chunk(self, r0, r1, dtype=torch.float32):
        """Rows [r0, r1) of this shard as a dense device matrix (deterministic in (seed, global rows))."""
        out = torch.empty(r1 - r0, self.m, dtype=dtype, device=self.device)
        for cell, a, b in self._cells(r0, r1):
            g = self._cell_rng(cell, 17)
            c0 = cell * self.CELL
            U = torch.empty(self.CELL, self.k, dtype=torch.float32, device=self.device).exponential_(1.0, generator=g)
            U = U[a - c0:b - c0] * self._u_scale(a - self.row0, b - self.row0)
            lam = U @ self.V.t()
            g2 = self._cell_rng(cell, 29)
            keep = torch.rand(self.CELL, self.m, generator=g2, device=self.device)[a - c0:b - c0] < self.pi_d
            out[a - self.row0 - r0:b - self.row0 - r0] = (torch.floor(lam) * keep).to(dtype)
        return out

    def initial_shapes(self):
        """`use_factors=False` start (gap.py:52, 62): a1 ~ Gamma(1) per local cell row, b1 ~ Gamma(1)
        per gene (replicated).  Returns float64 device tensors (n, k), (m, k)."""
        a1 = torch.empty(self.n, self.k, dtype=torch.float64, device=self.device)
        for cell, a, b in self._cells(0, self.n):
            g = self._cell_rng(cell, 43)
            c0 = cell * self.CELL
            e = torch.empty(self.CELL, self.k, dtype=torch.float64, device=self.device).exponential_(1.0, generator=g)
            a1[a - self.row0:b - self.row0] = e[a - c0:b - c0]
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed + 977)
        b1 = torch.empty(self.m, self.k, dtype=torch.float64, device=self.device).exponential_(1.0, generator=g)
        return a1, b1

    def labels(self, r0, r1):
        rows = self._group_rows(r0, r1)
        lab = torch.zeros(r1 - r0, dtype=torch.int64, device=self.device)
        for grp in range(self.n_groups):
            lab[(rows >= self.u_row_edges[grp]) & (rows < self.u_row_edges[grp + 1])] = grp
        return lab

# -*- coding: utf-8 -*-
"""Where the padded slots of the sliced layout sit (configs[3], the genes left after the hybrid layout's dense cut): slot
efficiency per gene tile (tiles in decreasing density) per 16-row slice, per wave (two slices) and with the tile barrier (eight
waves), and what a record granularity of 1 instead of 4 per row and iteration would give.   python tools/tile_slot_efficiency.py [rows]   (CPU)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd.singlecell.generation import SyntheticCounts   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, K, W = 30000, 100, 256
gen = SyntheticCounts(1000000, m, K, seed=1234 + 1000 * 4, device='cpu', zero_inflation_level=0.10, row0=0, n=n)
nz = gen.chunk(0, n).numpy() != 0
cnt = nz.sum(0)
order = np.argsort(-cnt, kind='stable')
gd = int((cnt >= 0.2 * n).sum()) // 32 * 32
nzs = nz[:, order[gd:]]
nt = (nzs.shape[1] + W - 1) // W
C = np.stack([nzs[:, t * W:(t + 1) * W].sum(1) for t in range(nt)], axis=1)
nb = n // 256
sl = C.reshape(nb, 16, 16, nt).max(2)                       # [block][slice][tile]: longest row of the slice
for gran in (4, 2, 1):
    it = (sl + gran - 1) // gran * gran                     # records per row, rounded to the iteration's granularity
    wave = it.reshape(nb, 8, 2, nt).max(2)
    grp = wave.max(1)
    nnz_t = C.sum(0)
    s_t, w_t, g_t = it.sum((0, 1)) * 16, wave.sum((0, 1)) * 32, grp.sum(0) * 256
    print('granularity %d: slot efficiency %.3f per slice, %.3f per wave, %.3f with the tile barrier' %
          (gran, nnz_t.sum() / s_t.sum(), nnz_t.sum() / w_t.sum(), nnz_t.sum() / g_t.sum()))
    if gran == 4:
        print('tile  density  %% of non-zeros  %% of issued slots  eff/slice  eff/wave  eff/barrier  iterations')
        for t in range(0, nt, 6):
            print('%4d  %.4f  %6.2f  %6.2f  %.3f  %.3f  %.3f  %5.2f' % (t, nnz_t[t] / (n * W), 100 * nnz_t[t] / nnz_t.sum(),
                  100 * g_t[t] / g_t.sum(), nnz_t[t] / s_t[t], nnz_t[t] / w_t[t], nnz_t[t] / g_t[t], grp[:, t].mean() / 4))
        lo = nnz_t / (n * W) < 0.04
        print('tiles below density 0.04: %d of %d, %.1f %% of the non-zeros, %.1f %% of the issued slots' %
              (lo.sum(), nt, 100 * nnz_t[lo].sum() / nnz_t.sum(), 100 * g_t[lo].sum() / g_t.sum()))

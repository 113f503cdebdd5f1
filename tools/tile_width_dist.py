# -*- coding: utf-8 -*-
"""Row-length distribution of the benchmark matrix for tools/ubench/tile_width.hip: for gene tiles of
W = 256 / 192 / 128 genes, the iterations (of 4 records per row) every wave of a 256-row block needs in every gene tile of the
SLICED part of configs[3] -- genes in decreasing density, the genes expressed in >= 20 % of the cells taken out as the hybrid
layout's dense block does -- as the K = 100 row kernel walks it: a wave = two 16-row slices, each padded to its longest row
(rounded up to 4 records), the wave runs the longer of the two.   python tools/tile_width_dist.py OUTDIR [rows]   (CPU)
File format (little endian): int32 W, nblocks, ntiles, 0; float64 nnz, slots issued per slice, per wave, per work-group
(all for the sample); int32 [nblocks][ntiles][8] iterations."""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd.singlecell.generation import SyntheticCounts   # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else '.'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
m, K = 30000, 100
os.makedirs(out, exist_ok=True)
gen = SyntheticCounts(1000000, m, K, seed=1234 + 1000 * 4, device='cpu', zero_inflation_level=0.10, row0=0, n=n)
nz = gen.chunk(0, n).numpy() != 0
cnt = nz.sum(0)
order = np.argsort(-cnt, kind='stable')
gd = int((cnt >= 0.2 * n).sum()) // 32 * 32
nz = nz[:, order[gd:]]
ms = nz.shape[1]
print('density %.4f, dense genes %d (%.1f %% of the non-zeros), sliced genes %d' % (float(cnt.sum()) / (n * m), gd,
      100.0 * cnt[order[:gd]].sum() / cnt.sum(), ms))
for W in (256, 192, 128):
    ntiles = (ms + W - 1) // W
    C = np.zeros((n, ntiles), np.int64)
    for t in range(ntiles):
        C[:, t] = nz[:, t * W:(t + 1) * W].sum(1)
    nb = n // 256
    sl = C.reshape(nb, 16, 16, ntiles).max(2)                 # [block][slice][tile]: longest row of the slice
    it = (sl + 3) // 4                                        # iterations of 4 records per row
    wave = it.reshape(nb, 8, 2, ntiles).max(2)                # [block][wave][tile]
    wit = np.ascontiguousarray(wave.transpose(0, 2, 1).astype(np.int32))      # [block][tile][wave]
    nnz = float(C.sum())
    slots = float(it.sum() * 4 * 16)
    wslots = float(wave.sum() * 4 * 32)
    gslots = float(wave.max(1).sum() * 4 * 256)
    with open(os.path.join(out, 'dist_%d.bin' % W), 'wb') as f:
        f.write(struct.pack('<4i4d', W, nb, ntiles, 0, nnz, slots, wslots, gslots))
        f.write(wit.tobytes())
    print('W = %3d: %4d tiles, slot efficiency %.3f per slice, %.3f per wave, %.3f with the tile barrier; mean iterations per tile %.2f'
          % (W, ntiles, nnz / slots, nnz / wslots, nnz / gslots, wave.mean()))

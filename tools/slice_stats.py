import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine as eng
from oriana_amd.singlecell import SyntheticCounts
# lock-step statistics of the sliced layout on one rank's share of C4 (131072 x 30000, z = 0.1)
n, m, K, z = 32768, 30000, 100, 0.1
gen = SyntheticCounts(n, m, K, seed=5234, device='cuda', zero_inflation_level=z)
ct = eng.CountTiles.from_chunks(n, m, gen.chunk, 8192, 'cuda')
for side, sl in (('row', ct.rslice), ('col', ct.cslice)):
    s = sl.cpu().numpy().astype(np.int64).reshape(ct.nrb, ct.ncb, 17)
    it = np.diff(s, axis=2)[:, :, :16] // 64          # iterations per slice (col side: last slice excludes dummy)
    if side == 'col':
        it = (np.diff(s, axis=2)[:, :, :16]) // 64
    mean = it.mean(axis=2); mx = it.max(axis=2)
    print('%s side: iterations per slice and tile: mean %.2f, max over the 16 slices %.2f -> barrier efficiency %.3f'
          % (side, mean.mean(), mx.mean(), mean.sum() / mx.sum()))
    # by column block (gene density decreases with cb)
    for cb in (0, 10, 30, 60, 90, 117):
        if cb < ct.ncb:
            print('   cb=%3d: mean %.2f max %.2f eff %.3f' % (cb, mean[:, cb].mean(), mx[:, cb].mean(), mean[:, cb].sum() / max(mx[:, cb].sum(), 1)))
print('slot efficiency', ct.slot_efficiency())

"""Slot efficiency of the packed layout on the generator's data (CPU, no GPU): the shipped scheme (16-row slices of fixed
rows, padded per tile; work-group waits for its slowest slice) against per-tile regrouping of the rows (sorted by their
count in the tile) -- DESIGN_HISTORY.md section 10(e) -- and the share of non-zeros / lane time per density band of the genes."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oriana_amd.singlecell.generation import SyntheticCounts
n, m, K = 4096, 30000, 100
gen = SyntheticCounts(1000000, m, K, seed=5234, device='cpu', zero_inflation_level=0.10, row0=0, n=n)
X = gen.chunk(0, n).numpy()
nz = X != 0
print('density', nz.mean())
order = np.argsort(-nz.sum(0), kind='stable')
nz = nz[:, order]
ncb = (m + 255) // 256
def tile_counts(rows):   # counts per (row, col tile)
    c = np.zeros((rows.shape[0], ncb), np.int64)
    for cb in range(ncb):
        c[:, cb] = rows[:, cb*256:(cb+1)*256].sum(1)
    return c
C = tile_counts(nz)           # [n, ncb]
nnz = C.sum()
def ceil4(x): return (x + 3) // 4 * 4
# (a) current: 256-row blocks, 16-row slices, per tile; WG time = max over slices
def scheme_fixed(C, blk, grp):
    slots = 0; wg = 0
    for b0 in range(0, C.shape[0], blk):
        Cb = C[b0:b0+blk]
        g = Cb.reshape(blk // grp, grp, -1).max(1)      # [groups, ncb]
        it = ceil4(g)
        slots += (it * grp).sum()
        wg += (it.max(0) * blk).sum()                   # lane-time incl. barrier wait
    return slots, wg
for blk, grp in ((256, 16), (256, 32)):
    s, w = scheme_fixed(C, blk, grp)
    print('fixed blk=%d grp=%d: slot eff %.3f  with barrier %.3f' % (blk, grp, nnz / s, nnz / w))
# (b) regrouped per tile: rows of a block sorted by their count in the tile, consecutive groups; wave w takes groups (w, G-1-w)
def scheme_regroup(C, blk, grp):
    slots = 0; wg = 0
    for b0 in range(0, C.shape[0], blk):
        Cb = np.sort(C[b0:b0+blk], axis=0)[::-1]        # sorted per tile (column-wise independent sort)
        g = Cb.reshape(blk // grp, grp, -1).max(1)
        it = ceil4(g)
        slots += (it * grp).sum()
        ng = it.shape[0]
        pair = it[:ng // 2] + it[ng // 2:][::-1]        # wave w: groups w and ng-1-w
        wg += (pair.max(0) * (blk // 2) ).sum() 
    return slots, wg
for blk, grp in ((512, 16), (512, 32), (1024, 32), (256, 16)):
    s, w = scheme_regroup(C, blk, grp)
    print('regroup blk=%d grp=%d: slot eff %.3f  with barrier %.3f' % (blk, grp, nnz / s, nnz / w))
# per-density breakdown of the current scheme
s_t = np.zeros(ncb); n_t = np.zeros(ncb)
for b0 in range(0, n, 256):
    Cb = C[b0:b0+256]
    it = ceil4(Cb.reshape(16, 16, -1).max(1))
    s_t += (it.max(0) * 256); n_t += Cb.sum(0)
for a in range(0, ncb, 12):
    print('tiles %3d-%3d: density %.3f  nnz share %.3f  lane-time share %.3f  eff %.3f' % (a, min(a+12, ncb)-1, n_t[a:a+12].sum() / (n * 256 * min(12, ncb - a)), n_t[a:a+12].sum() / nnz, s_t[a:a+12].sum() / s_t.sum(), n_t[a:a+12].sum() / s_t[a:a+12].sum()))

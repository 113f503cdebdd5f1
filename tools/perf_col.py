import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine as eng
from oriana_amd._lib import call, ptr, stream_ptr
from oriana_amd.singlecell import SyntheticCounts
n, m, K, z = 131072, 30000, 100, 0.1
dev = 'cuda'
gen = SyntheticCounts(n, m, K, seed=1234, device=dev, zero_inflation_level=z)
ct = eng.CountTiles.from_chunks(n, m, lambda a, b: gen.chunk(a, b), 8192, dev)
ws = eng.ZWorkspace(ct, K)
ws.FU.uniform_(0.1, 1.0); ws.s_cs.uniform_(0.0, 1.0)
nrb, ncb = ct.nrb, ct.ncb
coff = ct.coff.cpu().numpy()
per_tile = (coff[1:] - coff[:-1]).reshape(nrb, ncb)
per_cb = per_tile.sum(0)
print('nrb', nrb, 'ncb', ncb, 'slots per cb: max %.3g mean %.3g min %.3g' % (per_cb.max(), per_cb.mean(), per_cb.min()))
def run(items, reps=3):
    w = torch.tensor(items, dtype=torch.int32, device=dev).contiguous()
    ws.C.zero_()
    ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        call('oriana_col_pass', ct.c_struct, ptr(ws.s_cs), ptr(ws.FU), ptr(ws.C), K, ptr(w), w.shape[0], stream_ptr())
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
cbm = int(np.argsort(per_cb)[len(per_cb) // 2])      # median-density column block
for cb in (0, cbm):
    band = 64
    slots = per_tile[:band, cb].sum()
    for nw in (1, 64, 256, 512, 1024):
        t = run([[cb, 0, band]] * nw)
        print('cb=%d (%.0f slots/WG, %.1f iters/tile/slice) nWG=%4d: %.3f ms -> %.1f ns per slice-iteration per wave' % (cb, slots, slots / band / 16 / 64, nw, t, t * 1e6 / (slots / 16 / 64)))
# the real work list
t = run(ct.col_work_for(K).cpu().numpy().tolist())
print('work list (%d items): %.3f ms' % (ct.col_work_for(K).shape[0], t))
w = ct.col_work_for(K).cpu().numpy()
work = np.array([per_tile[a:b, c].sum() for c, a, b in w])
print('item slots: max %.3g mean %.3g; sum/256 = %.3g -> ideal time at the 256-WG rate' % (work.max(), work.mean(), work.sum() / 256))
h = ct.host_arrays()
cs = h['cslice'].astype(np.int64)
nit = (cs[:, 1:] - cs[:, :-1]) // 64           # [ntiles, 16]
mx = nit.max(1); mean = nit.mean(1)
print('col side: sum(max niter) = %.4g, sum(mean niter) = %.4g, ratio %.3f; tiles %d' % (mx.sum(), mean.sum(), mx.sum() / mean.sum(), len(mx)))
print('predicted col time: %.2f ms (1.45us/iter + 3us/tile)' % ((mx.sum() * 1.45e-3 + len(mx) * 3e-3) / 256))
rs = h['rslice'].astype(np.int64)
nitr = (rs[:, 1:] - rs[:, :-1]) // 64
print('row side: sum(max niter) = %.4g, sum(mean niter) = %.4g, ratio %.3f' % (nitr.max(1).sum(), nitr.mean(1).sum(), nitr.max(1).sum() / nitr.mean(1).sum()))
d = per_tile.sum(0) / (nrb * 65536.0)
print('col block density quantiles', np.quantile(d, [0, .1, .25, .5, .75, .9, 1]).round(3))

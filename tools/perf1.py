import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine as eng
from oriana_amd._lib import call, ptr, stream_ptr
from oriana_amd.singlecell import SyntheticCounts
n = int(sys.argv[1]); m = int(sys.argv[2]); K = int(sys.argv[3]); z = float(sys.argv[4])
dev = 'cuda'
t0 = time.time()
gen = SyntheticCounts(n, m, K, seed=1234, device=dev, zero_inflation_level=z)
ct = eng.CountTiles.from_chunks(n, m, lambda a, b: gen.chunk(a, b), 8192, dev)
torch.cuda.synchronize()
print('packed n=%d m=%d nnz=%d (%.3f) in %.1fs; resident %.2f GB; slot eff row %.3f col %.3f' % ((n, m, ct.nnz, ct.nnz / (n * m), time.time() - t0, ct.bytes_resident() / 1e9) + ct.slot_efficiency()))
ws = eng.ZWorkspace(ct, K)
g = torch.Generator(device=dev); g.manual_seed(1)
a1 = torch.empty(n, K, device=dev, dtype=torch.float64).exponential_(1.0, generator=g).clamp_min(1e-15)
b1 = torch.empty(m, K, device=dev, dtype=torch.float64).exponential_(1.0, generator=g).clamp_min(1e-15)
lu = torch.digamma(a1.float().double()).float().contiguous(); lv = torch.digamma(b1.float().double()).float().contiguous()
Zi = torch.empty(n, K, device=dev); Zj = torch.empty(m, K, device=dev)
def ev(): return torch.cuda.Event(enable_timing=True)
st = stream_ptr()
for it in range(3):
    evs = [ev() for _ in range(8)]
    evs[0].record()
    eng.factor_prep(ws.FU, lu); eng.factor_prep(ws.FV, lv, row_index=ct.col_perm)
    Zi.zero_(); Zj.zero_(); ws.C.zero_(); ws.tile_flag.zero_()
    evs[1].record()
    call('oriana_row_pass', ct.c_struct, ptr(ws.FU), ptr(ws.FV), None, ptr(ws.R), ptr(ws.s_cs), None, None, ptr(ws.tile_flag), K, st)
    evs[2].record()
    call('oriana_fixup', ct.c_struct, ptr(ws.tile_flag), ptr(ws.s_cs), None, None, ptr(lu), ptr(lv), None, None, None, None, ptr(Zi), ptr(Zj), None, K, 0, st)
    evs[3].record()
    eng.col_pass(ct, ws.s_cs, ws.FU, ws.C, K)
    evs[4].record()
    call('oriana_finalize', ptr(Zi), ptr(ws.FU), ptr(ws.R), None, None, n, K, 1, st)
    call('oriana_finalize', ptr(Zj), ptr(ws.FV), ptr(ws.C), None, ptr(ct.col_perm), m, K, 1, st)
    evs[5].record()
    torch.cuda.synchronize()
    names = ['prep+zero', 'row_pass', 'fixup', 'col_pass', 'finalize']
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(5)]
    tot = sum(ts)
    alg = 4.0 * n * m + 4.0 * K * (2 * n + 2 * m)
    print(' '.join('%s=%.3fms' % (a, b) for a, b in zip(names, ts)), 'total=%.3fms' % tot, 'alg GB/s=%.0f' % (alg / tot / 1e6),
          'clk/nnz row=%.2f col=%.2f' % (ts[1] * 1e-3 * 2.4e9 * 256 / ct.nnz, ts[3] * 1e-3 * 2.4e9 * 256 / ct.nnz))
print('flags', int(ws.tile_flag.sum()), 'rowsum check', float((Zi.sum(1) - 0).abs().max()), float(Zi.sum()), float(Zj.sum()))

"""Column pass alone on one rank's share of C4 (125,000 x 30,000, K = 100) for several work-list sizes, plus the
deterministic (slab + ordered reduction) flush against the atomic one."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine as eng, _lib
from oriana_amd.singlecell import SyntheticCounts
n, m, K, z = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 0.1
dev = 'cuda'
gen = SyntheticCounts(n, m, K, seed=1234, device=dev, zero_inflation_level=z)
ct = eng.CountTiles.from_chunks(n, m, lambda a, b: gen.chunk(a, b), 8192, dev)
ws = eng.ZWorkspace(ct, K)
g = torch.Generator(device=dev); g.manual_seed(1)
a1 = torch.empty(n, K, device=dev, dtype=torch.float64).exponential_(1.0, generator=g).clamp_min(1e-15)
b1 = torch.empty(m, K, device=dev, dtype=torch.float64).exponential_(1.0, generator=g).clamp_min(1e-15)
lu = torch.digamma(a1.float().double()).float().contiguous(); lv = torch.digamma(b1.float().double()).float().contiguous()
Zi = torch.empty(n, K, device=dev); Zj = torch.empty(m, K, device=dev)
eng.zq_gap(ws, Zi, Zj, lu, lv)
w = int(_lib.load().oriana_col_block_tiles(K))
def run(reps=5):
    ts = []
    for _ in range(reps):
        ws.C.zero_()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); eng.col_pass(ct, ws.s_cs, ws.FU, ws.C, K); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for T in [None] + [int(x) for x in os.environ.get("TARGETS", "576,1152,2304,4608,9216").split(",")]:
    ct._col_work[w] = ct._build_col_work(target_items=T, width=w)
    for det in ((False, True) if os.environ.get("DET", "1") == "1" else (False,)):
        eng.set_deterministic(det)
        print('items %5d (target %s) %s: %.3f ms' % (ct._col_work[w].shape[0], T, 'slabs+reduce' if det else 'atomics     ', run()))
eng.set_deterministic(False)

import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine as eng
from oriana_amd.singlecell import SyntheticCounts
# effect of ordering the cells by depth inside a packing chunk, on cells of heterogeneous depth:
# the synthetic matrix with every cell's entries thinned by its own keep probability in [0.15, 1]
n, m, K = int(os.environ.get("N", 65536)), 30000, 100
gen = SyntheticCounts(n, m, K, seed=4234, device='cuda', zero_inflation_level=0.1)
def chunk(r0, r1):
    X = gen.chunk(r0, r1)
    g = torch.Generator(device='cuda'); g.manual_seed(1000 + r0)
    keep_p = 0.15 + 0.85 * torch.rand(r1 - r0, 1, generator=g, device='cuda')
    return X * (torch.rand(X.shape, generator=g, device='cuda') < keep_p)
for sort_rows in (False, True):
    ct = eng.CountTiles.from_chunks(n, m, chunk, 8192, 'cuda', sort_rows=sort_rows)
    ws = eng.ZWorkspace(ct, K)
    lu = torch.randn(n, K, device='cuda'); lv = torch.randn(m, K, device='cuda')
    Zi = torch.empty(n, K, device='cuda'); Zj = torch.empty(m, K, device='cuda')
    t = eng.KernelTimer(); ws.timer = t
    for _ in range(3):
        eng.zq_gap(ws, Zi, Zj, lu, lv)
    torch.cuda.synchronize()
    s = t.summary()
    print('sort_rows=%s nnz=%d slot efficiency (row, col) = %.3f %.3f  row pass %.2f ms  col pass %.2f ms'
          % (sort_rows, ct.nnz, *ct.slot_efficiency(), s['row_pass'][1], s['col_pass'][1]))
    del ct, ws

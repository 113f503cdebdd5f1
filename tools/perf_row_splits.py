# -*- coding: utf-8 -*-
"""Row pass of pCMF's loop nest against the gene-tile split of its row blocks (ORIANA_ROW_SPLITS=<forced whole-grid split>, ORIANA_ROW_SPLIT_ROUNDS=off:
no split of the last round; read once per process): python tools/perf_row_splits.py n m K [sweeps]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd import engine                              # noqa: E402
from oriana_amd.singlecell import SyntheticCounts          # noqa: E402

n, m, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device('cuda')
gen = SyntheticCounts(n, m, K, seed=77, device=dev, zero_inflation_level=0.1)
ct = engine.CountTiles.from_chunks(n, m, gen.chunk, 8192, dev, sort_rows=os.environ.get('SORT_ROWS') == '1',
                                   dense_density=float(os.environ['DENSE']) if os.environ.get('DENSE') else None)
ws = engine.ZWorkspace(ct, K)
lu = torch.randn(n, K, device=dev) * 0.3
lv = torch.randn(m, K, device=dev) * 0.3
Zi = torch.empty(n, K, device=dev); Zj = torch.empty(m, K, device=dev)
for _ in range(2):
    engine.zq_gap(ws, Zi, Zj, lu, lv)
ws.timer = engine.KernelTimer()
for _ in range(reps):
    engine.zq_gap(ws, Zi, Zj, lu, lv)
torch.cuda.synchronize()
s = ws.timer.summary()
sp = ws.row_split
print('slot efficiency', ct.slot_efficiency(), 'sort_rows', ct.sort_rows, 'gd', ct.gd)
print('n=%d m=%d K=%d col items %d row blocks %d whole %d parts %d edges %s: row pass %.3f ms, col pass %.3f ms; check %.6e' % (
    n, m, K, int(ct.col_work_for(K).shape[0]), ct.nrb, sp.nfull, sp.parts, list(sp.edge[:sp.parts + 1]), s['row_pass'][1], s['col_pass'][1], float(Zi.double().sum())))

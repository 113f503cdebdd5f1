# -*- coding: utf-8 -*-
"""ZI-pCMF at configs[2] (100k x 20k, K = 50): flagged tiles, rejected rows and the row maxima of E[log U], E[log V] over the sweeps,
then the pass timings of five more sweeps (DESIGN_HISTORY.md 10 m).  ORIANA_DEN_THRESHOLD=fixed: the constant threshold of round 3."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine
from oriana_amd.models import ZIGaP
from oriana_amd.singlecell import SyntheticCounts
n, m, K = 100000, 20000, 50
dev = torch.device('cuda')
gen = SyntheticCounts(n, m, K, seed=77, device=dev, zero_inflation_level=0.1)
ct = engine.CountTiles.from_chunks(n, m, gen.chunk, 8192, dev)
a1, b1 = gen.initial_shapes()
model = ZIGaP(ct, k=K, init=(os.environ['INIT'] if os.environ.get('INIT') else (a1, b1)), device=dev)       # INIT=nmf: the reference's default start
ws = model._ws
NS = int(os.environ.get('SWEEPS', '26'))
for it in range(NS):
    model.step()
    if it in (0, 1, 2, 5, 10, 15, 20, 25) or it % 10 == 5:
        torch.cuda.synchronize()
        fl = ws.tile_flag[:ct.nrb * ct.ncb]
        nanrows_u = int(torch.isnan(ws.FU).any(1).sum()) if False else -1
        fu_fill = int((ws.FU[:, 0] == ws.FU[:, 1]).logical_and(ws.FU[:, 0] < 1e-30).logical_and(ws.FU[:, 0] > 0).sum())
        fv_fill = int((ws.FV[:, 0] == ws.FV[:, 1]).logical_and(ws.FV[:, 0] < 1e-30).logical_and(ws.FV[:, 0] > 0).sum())
        lu, lv = model._log_U_hat, model._log_V_hat
        fin_u = torch.isfinite(lu).all(1) & (lu.max(1).values > -1e10); fin_v = torch.isfinite(lv).all(1) & (lv.max(1).values > -1e10)
        print('   stats: cu %.2f cv %.2f | rows with max < -1e10: U %d V %d | U row max quantiles (1%%, 50%%, 99%%) %s | V %s | thr %.3g' % (
            float(ws.stats[0] / max(float(ws.stats[2]), 1)), float(ws.stats[3] / max(float(ws.stats[5]), 1)), int((~fin_u).sum()), int((~fin_v).sum()),
            [round(float(v), 1) for v in torch.quantile(lu.max(1).values[fin_u].double(), torch.tensor([0.01, 0.5, 0.99], dtype=torch.float64, device=lu.device))],
            [round(float(v), 1) for v in torch.quantile(lv.max(1).values[fin_v].double(), torch.tensor([0.01, 0.5, 0.99], dtype=torch.float64, device=lu.device))],
            float(ws.stats[7])))
        print('sweep %d: flagged tiles %d of %d; FILL rows U %d V %d; row max logU mean %.2f sd %.2f min %.2f max %.2f; logV mean %.2f sd %.2f min %.2f max %.2f' % (
            it, int(fl.sum()), fl.numel(), fu_fill, fv_fill, float(lu.max(1).values.mean()), float(lu.max(1).values.std()), float(lu.max(1).values.min()), float(lu.max(1).values.max()),
            float(lv.max(1).values.mean()), float(lv.max(1).values.std()), float(lv.max(1).values.min()), float(lv.max(1).values.max())))
ws.timer = engine.KernelTimer()
for _ in range(5):
    model.step()
torch.cuda.synchronize()
print({k: round(v[1], 3) for k, v in ws.timer.summary().items()})

"""Hybrid layout: time of the responsibility pass against the density threshold of the dense-gene block.
   python tools/perf_dense_pass.py n m K z [thresholds...]   (c4-like synthetic counts, bench.py's generator)"""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine as eng
from oriana_amd.singlecell import SyntheticCounts
n = int(sys.argv[1]); m = int(sys.argv[2]); K = int(sys.argv[3]); z = float(sys.argv[4])
ths = [float(a) for a in sys.argv[5:]] or [0.0, 0.3, 0.2, 0.15, 0.1]
dev = 'cuda'
gen = SyntheticCounts(n, m, K, seed=5234, device=dev, zero_inflation_level=z)
g = torch.Generator(device=dev); g.manual_seed(1)
a1 = torch.empty(n, K, device=dev, dtype=torch.float64).exponential_(1.0, generator=g).clamp_min(1e-15)
b1 = torch.empty(m, K, device=dev, dtype=torch.float64).exponential_(1.0, generator=g).clamp_min(1e-15)
lu = torch.digamma(a1.float().double()).float().contiguous(); lv = torch.digamma(b1.float().double()).float().contiguous()
del a1, b1
ref = None
for th in ths:
    t0 = time.time()
    ct = eng.CountTiles.from_chunks(n, m, lambda a, b: gen.chunk(a, b), 8192, dev, dense_density=(th or None))
    ws = eng.ZWorkspace(ct, K)
    torch.cuda.synchronize()
    setup = time.time() - t0
    Zi = torch.empty(n, K, device=dev); Zj = torch.empty(m, K, device=dev)
    for _ in range(2):
        eng.zq_gap(ws, Zi, Zj, lu, lv)
    ws.timer = eng.KernelTimer()
    reps = 5
    for _ in range(reps):
        eng.zq_gap(ws, Zi, Zj, lu, lv)
    torch.cuda.synchronize()
    ks = {k: round(v[1] * v[0] / reps, 3) for k, v in ws.timer.summary().items()}
    tot = sum(ks.values())
    dn_entries = ct.n * ct.gd
    dn_nnz = ct.dense.nnz if ct.dense is not None else 0
    out = dict(threshold=th, gd=ct.gd, dense_entries=dn_entries, dense_nnz=dn_nnz, nnz=ct.nnz, setup_s=round(setup, 1),
               ms=ks, pass_ms=round(tot, 3), mem_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1),
               flags=int(ws.tile_flag.sum()) + (int(ws.dn_flag.sum()) if ct.dense is not None else 0))
    if ct.dense is not None:
        dms = ks.get('dense_row', 0) + ks.get('dense_col', 0) + ks.get('dense_images', 0)
        out['dense_ps_per_entry'] = round(dms * 1e9 / max(dn_entries, 1), 3)
        out['dense_ps_per_nnz'] = round(dms * 1e9 / max(dn_nnz, 1), 3)
    sp = ks.get('row_pass', 0) + ks.get('col_pass', 0)
    out['sparse_ps_per_nnz'] = round(sp * 1e9 / max(ct.nnz_sparse, 1), 3)
    zi, zj = Zi.double(), Zj.double()
    if ref is None:
        ref = (zi.clone(), zj.clone())
    else:
        out['vs_first'] = [float(((zi - ref[0]).abs() / (ref[0].abs() + ref[0].abs().amax(0, keepdim=True) + 1e-300)).max()),
                           float(((zj - ref[1]).abs() / (ref[1].abs() + ref[1].abs().amax(0, keepdim=True) + 1e-300)).max())]
    print(json.dumps(out), flush=True)
    del ct, ws, Zi, Zj
    torch.cuda.empty_cache()

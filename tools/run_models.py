import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine
from oriana_amd.models import GaP, ZIGaP, SparseGaP, SparseZIGaP
from oriana_amd.singlecell import SyntheticCounts
name, n, m, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device('cuda')
gen = SyntheticCounts(n, m, K, seed=77, device=dev, zero_inflation_level=0.1)
t0 = time.time()
ct = engine.CountTiles.from_chunks(n, m, gen.chunk, 8192, dev)
a1, b1 = gen.initial_shapes()
cls = dict(GaP=GaP, ZIGaP=ZIGaP, SparseGaP=SparseGaP, SparseZIGaP=SparseZIGaP)[name]
model = cls(ct, k=K, init=(a1, b1), device=dev)
torch.cuda.synchronize(); print('setup %.1fs nnz=%d mem=%.1f GB' % (time.time() - t0, ct.nnz, torch.cuda.max_memory_allocated() / 1e9))
sweeps = int(os.environ.get('SWEEPS', '4'))           # (long runs: every 10th sweep is printed)
for it in range(sweeps):
    torch.cuda.synchronize(); t0 = time.time()
    model.step()
    torch.cuda.synchronize()
    if sweeps <= 8 or it < 3 or it % 10 == 9:
        print('sweep %d: %.1f ms' % (it, (time.time() - t0) * 1e3))
st_alpha = model.alpha1.asarray()
print('alpha1[:4]', st_alpha[:4], 'finite', np.isfinite(st_alpha).all(), 'mem=%.1f GB' % (torch.cuda.max_memory_allocated() / 1e9))
if hasattr(model, 'pi_d'): print('pi_d mean', float(model.pi_d.tensor.mean()))
if hasattr(model, 'pi_s'): print('pi_s mean', float(model.pi_s.tensor.mean()))

import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oriana_amd import engine
from oriana_amd.models import GaP
from oriana_amd.singlecell import SyntheticCounts
n, m, K = 10000, 2000, 20
gen = SyntheticCounts(n, m, K, seed=5234, device='cuda', zero_inflation_level=0.1)
ct = engine.CountTiles.from_chunks(n, m, gen.chunk, 8192, 'cuda')
a1, b1 = gen.initial_shapes()
for mode in ('eager', 'graph'):
    model = GaP(ct, k=K, init=(a1, b1))
    if mode == 'graph': model.capture_graph()
    for _ in range(5): model.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): model.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, '%.3f ms/sweep' % (dt / 200 * 1e3), float(model.alpha1.tensor.sum()))

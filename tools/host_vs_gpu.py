# -*- coding: utf-8 -*-
"""Is an eager configs[1] sweep bound by the host (launch path) or by the GPU?  Wall time of the launch loop alone
against the time until the GPU has drained, plus a cProfile of the launch loop.   python tools/host_vs_gpu.py [--profile]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd.models import GaP   # noqa: E402


def main():
    rng = np.random.default_rng(0)
    n, m, K = 10000, 2000, 20
    X = ((rng.poisson(3.0, size=(n, m)) + 1) * (rng.random((n, m)) < 0.1)).astype(np.float32)
    a1 = rng.gamma(1.0, size=(n, K)); b1 = rng.gamma(1.0, size=(m, K))
    model = GaP(X, k=K, init=(a1, b1), device='cuda')
    for _ in range(20):
        model.step()
    torch.cuda.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        model.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('launch loop %.1f us per sweep, drained after %.1f us per sweep' % ((t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6))
    if '--profile' in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(reps):
            model.step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)


if __name__ == '__main__':
    main()

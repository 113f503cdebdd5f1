# -*- coding: utf-8 -*-
"""The two dense contractions of a ZI sweep (D update + next D_hat V; D_hat^T U) per arithmetic: ms and ps per entry.
   python tools/perf_zi_per_k.py n m K [K ...]        (GPU)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd import _lib                         # noqa: E402
from oriana_amd._lib import call, ptr, stream_ptr   # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(1)
    X = (torch.rand(n, m, device=dev, generator=g) < 0.1).float()
    mask = torch.zeros(((n + 31) // 32) * m, dtype=torch.int32, device=dev)
    call('oriana_nzmask_f32', ptr(mask), ptr(X), n, m, stream_ptr())
    D = X                                                  # (overwritten by the sweep: the non-zeros stay 1)
    from oriana_amd import engine
    tiles = engine.nzmask_tiles(mask, n, m)
    lib = _lib.load()
    for K in [int(a) for a in sys.argv[3:]]:
        U = torch.rand(n, K, dtype=torch.float64, device=dev, generator=g) * 0.2
        V = torch.rand(m, K, dtype=torch.float64, device=dev, generator=g) * 0.2
        pi = torch.rand(m, dtype=torch.float64, device=dev, generator=g)
        cs = torch.zeros(m, dtype=torch.float64, device=dev)
        DV = torch.zeros(n, K, dtype=torch.float64, device=dev)
        out = torch.zeros(m, K, dtype=torch.float64, device=dev)
        lgs = torch.zeros(int(lib.oriana_dropout_sweep_scratch_floats(m, K)), dtype=torch.float32, device=dev)
        dts = torch.zeros(int(lib.oriana_dense_t_scratch_floats(n, K)), dtype=torch.float32, device=dev)
        res = []
        for arith, name in ((1, 'bf16x3'), (0, 'f32')):
            ts = timed(lambda: call('oriana_dropout_sweep_fused_tiles', ptr(D), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(tiles), ptr(cs),
                                    ptr(V), ptr(DV), ptr(lgs), arith, n, m, K, stream_ptr()))
            td = timed(lambda: call('oriana_dense_t_times_factor_f32', ptr(out), ptr(D), ptr(U), ptr(dts), arith, n, m, K,
                                    stream_ptr()))
            res.append('%s: D update %.2f ms (%.2f ps/entry), D^T U %.2f ms (%.2f ps/entry)' % (
                name, ts, ts * 1e9 / (n * m), td, td * 1e9 / (n * m)))
        print('n=%d m=%d K=%d | %s' % (n, m, K, ' | '.join(res)), flush=True)


if __name__ == '__main__':
    main()

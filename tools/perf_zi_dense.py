# -*- coding: utf-8 -*-
"""Times the dense kernels of a ZI sweep on synthetic operands: the float32 matrix-core path (csrc/dense_f32.hip)
next to the float64 one (csrc/dense_mfma.hip).   python3 tools/perf_zi_dense.py [n m K]   (default: configs[2])"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd._lib import call, ptr, stream_ptr   # noqa: E402

n, m, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (100000, 20000, 50)
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(1)
D = torch.rand(n, m, generator=g, device=dev)
U = torch.rand(n, K, generator=g, device=dev, dtype=torch.float64) * 0.5
V = torch.rand(m, K, generator=g, device=dev, dtype=torch.float64) * 0.5
pi = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
mask = torch.zeros(((n + 31) // 32) * m, dtype=torch.int32, device=dev)
call('oriana_nzmask_f32', ptr(mask), ptr((D < 0.1).float()), n, m, stream_ptr())
cs = torch.zeros(m, dtype=torch.float64, device=dev)
o1 = torch.zeros(n, K, dtype=torch.float64, device=dev)
o2 = torch.zeros(m, K, dtype=torch.float64, device=dev)
from oriana_amd import _lib   # noqa: E402
lgs = torch.zeros(int(_lib.load().oriana_dropout_sweep_scratch_floats(m, K)), dtype=torch.float32, device=dev)


def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


st = stream_ptr()
rows = []
# [r6] the sweep entry with the per-lane flags (csrc/dense_zi.hip); a round-5 library (ORIANA_HIP_LIB=...) has the old entry only
if hasattr(_lib.load(), 'oriana_nzmask_tiles'):
    from oriana_amd import engine
    tiles = engine.nzmask_tiles(mask, n, m)
    def sweep(arith, Vn, DV):
        call('oriana_dropout_sweep_fused_tiles', ptr(D), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(tiles), ptr(cs), ptr(Vn), ptr(DV),
             ptr(lgs), arith, n, m, K, st)
else:
    def sweep(arith, Vn, DV):
        call('oriana_dropout_sweep_fused', ptr(D), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(cs), ptr(Vn), ptr(DV), ptr(lgs), arith, n, m, K, st)
if K <= 100:
    t = timeit(lambda: sweep(1, V, o1))
    rows.append(('bf16x3  D update + D V_next (fused)', t, 2.0 * n * m * (K + 32 * ((K + 31) // 32)) / t / 1e9, 4.0 * n * m / t / 1e9))
t = timeit(lambda: sweep(0, V, o1))
rows.append(('f32  D update + D V_next (fused)', t, 2.0 * n * m * (K + 32 * ((K + 31) // 32)) / t / 1e9, 4.0 * n * m / t / 1e9))
if '--shipped-only' not in sys.argv:       # (counter runs: only the two kernels a sweep launches)
    t = timeit(lambda: sweep(0, None, None))
    rows.append(('f32  D update alone', t, 2.0 * n * m * K / t / 1e9, 4.0 * n * m / t / 1e9))
scr = torch.zeros(int(_lib.load().oriana_dense_t_scratch_floats(n, K)), dtype=torch.float32, device=dev)
if K <= 100:
    t = timeit(lambda: call('oriana_dense_t_times_factor_f32', ptr(o2), ptr(D), ptr(U), ptr(scr), 1, n, m, K, st))
    rows.append(('bf16x3  D^T U', t, 2.0 * n * m * 32 * ((K + 31) // 32) / t / 1e9, 4.0 * n * m / t / 1e9))
t = timeit(lambda: call('oriana_dense_t_times_factor_f32', ptr(o2), ptr(D), ptr(U), ptr(scr), 0, n, m, K, st))
rows.append(('f32  D^T U', t, 2.0 * n * m * 32 * ((K + 31) // 32) / t / 1e9, 4.0 * n * m / t / 1e9))
if '--f64' in sys.argv:
    t = timeit(lambda: call('oriana_dropout_update_fused', None, ptr(D), ptr(U), ptr(V), ptr(pi), ptr(mask), ptr(cs), n, m, K, st))
    rows.append(('f64  D update', t, 2.0 * n * m * K / t / 1e9, 4.0 * n * m / t / 1e9))
    t = timeit(lambda: call('oriana_dense_times_factor', ptr(o1), ptr(D), ptr(V), n, m, K, 0, st))
    rows.append(('f64  D V', t, 2.0 * n * m * K / t / 1e9, 4.0 * n * m / t / 1e9))
    t = timeit(lambda: call('oriana_dense_times_factor', ptr(o2), ptr(D), ptr(U), n, m, K, 1, st))
    rows.append(('f64  D^T U', t, 2.0 * n * m * K / t / 1e9, 4.0 * n * m / t / 1e9))
print('%d x %d, K = %d' % (n, m, K))
for name, t, tf, tb in rows:
    print('%-36s %7.2f ms   %6.1f TFLOP/s issued   D_hat traffic %.2f TB/s' % (name, t, tf, tb))

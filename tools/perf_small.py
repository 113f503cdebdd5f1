# -*- coding: utf-8 -*-
"""Where a configs[1]-sized row pass spends its time: fixed cost vs per-tile cost (GPU).
   python tools/perf_small.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd import engine                      # noqa: E402
from oriana_amd._lib import call, ptr, stream_ptr   # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3        # us


def main():
    rng = np.random.default_rng(0)
    n, K = 10000, int(os.environ.get('K', 20))
    for m, dens in ((2000, 0.1), (2000, 0.001), (256, 0.1), (256, 0.001)):
        X = (rng.poisson(3.0, size=(n, m)) + 1) * (rng.random((n, m)) < dens)
        ct = engine.CountTiles.from_dense(X.astype(np.float32), 'cuda')
        ws = engine.ZWorkspace(ct, K)
        lu = torch.randn(n, K, device='cuda'); lv = torch.randn(m, K, device='cuda')
        engine.factor_prep_pair(ws, lu, lv)
        st = stream_ptr()
        out = []
        slabs = torch.zeros(ct.ncb, n, ws.Kp, device='cuda')
        for gs in sorted(g for g in {1, 2, 4, ct.ncb} if g <= ct.ncb):
            t = timed(lambda: call('oriana_row_pass_split', ct.sparse_struct, ptr(ws.FU), ptr(ws.FV), ptr(slabs), ptr(ws.s_cs),
                                   ptr(ws.tile_flag), K, gs, st))
            out.append('gs=%d %.1f us' % (gs, t))
        tc = timed(lambda: engine.col_pass(ct, ws.s_cs, ws.FU, ws.C, K))
        tz = timed(lambda: ws.tile_flag.zero_())
        tp = timed(lambda: engine.factor_prep_pair(ws, lu, lv))
        print('m=%d density=%.3f nnz=%d ncb=%d: row %s | col %.1f us | fill %.1f us | prep pair %.1f us' % (m, dens, ct.nnz, ct.ncb, ', '.join(out), tc, tz, tp))


if __name__ == '__main__':
    main()

# -*- coding: utf-8 -*-
"""Time oriana_gamma_update_finalize / oriana_mstep_gamma_pair at a configs[1] size (GPU).
   ORIANA_GU_RPB=<rows per block> python tools/perf_gamma.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oriana_amd import engine                      # noqa: E402
from oriana_amd._lib import call, ptr, stream_ptr   # noqa: E402
from perf_small import timed                        # noqa: E402


def main():
    dev = 'cuda'
    shapes = ((10000, 20), (2000, 20), (125000, 100), (30000, 100), (1000000, 100))
    if os.environ.get('SHAPES'):                 # e.g. SHAPES=1000000x96,1000000x128
        shapes = tuple(tuple(int(v) for v in t.split('x')) for t in os.environ['SHAPES'].split(','))
    for r, K in shapes:
        Kp = engine.kpad(K)
        # [r6] shapes as a sweep sees them: most of a row's K shapes stay near the prior (a1 = prior + F R below 6 -- the regime in
        # which the digamma's recurrence runs; round 5's all-large shapes hid a data-dependent loop of float64 divisions there),
        # a tenth carries the counts.  SMALL_SHARE=0 restores the old input.
        F = torch.rand(r, Kp, device=dev); R = torch.rand(r, Kp, device=dev) * 50
        R = R * (torch.rand(r, Kp, device=dev) >= float(os.environ.get('SMALL_SHARE', '0.9')))
        Z = torch.zeros(r, K, device=dev)
        idx = torch.randperm(r, device=dev).to(torch.int32)
        p1 = torch.rand(K, dtype=torch.float64, device=dev) + 0.5
        p2 = torch.rand(K, dtype=torch.float64, device=dev) + 0.5
        rate = torch.rand(K, dtype=torch.float64, device=dev) * 100
        a1, a2, E = (torch.empty(r, K, dtype=torch.float64, device=dev) for _ in range(3))
        El = torch.empty(r, K, device=dev)
        sums = torch.zeros(2, K, dtype=torch.float64, device=dev)
        st = stream_ptr()
        t = timed(lambda: call('oriana_gamma_update_finalize', ptr(a1), ptr(a2), ptr(E), ptr(El), ptr(sums[0]), ptr(sums[1]),
                               ptr(p1), ptr(p2), ptr(Z), ptr(F), ptr(R), 1, ptr(idx), ptr(rate), r, K, st))
        nblk = int(__import__('oriana_amd')._lib.load().oriana_gamma_update_prep_blocks(r, K))
        tp = float('nan')
        if nblk:
            FUn = torch.zeros(r, Kp, device=dev); mu = torch.zeros(r, device=dev); up = torch.zeros(4 * nblk, device=dev)
            tp = timed(lambda: call('oriana_gamma_update_finalize_prep', ptr(a1), ptr(a2), ptr(E), ptr(El), ptr(sums[0]), ptr(sums[1]),
                                    ptr(p1), ptr(p2), ptr(Z), ptr(F), ptr(R), 1, 0, None, ptr(rate), r, K, ptr(FUn), ptr(mu), ptr(up), st))
            tn = timed(lambda: call('oriana_gamma_update_finalize_prep', ptr(a1), ptr(a2), ptr(E), ptr(El), ptr(sums[0]), ptr(sums[1]),
                                    ptr(p1), ptr(p2), ptr(Z), ptr(F), ptr(R), 1, 0, None, ptr(rate), r, K, None, None, None, st))
            print('   no row permutation: %.1f us (%.2f TB/s at 44 B); with the next sweep\'s FU, row maxima, statistics: %.1f us (%.2f TB/s at 48 B)' % (
                tn, 44.0 * r * K / (tn * 1e-6) / 1e12, tp, 48.0 * r * K / (tp * 1e-6) / 1e12))
        q = [p1.clone(), p2.clone(), p1.clone(), p2.clone()]
        keep = torch.zeros(2, K, dtype=torch.float64, device=dev)
        sE = torch.rand(K, dtype=torch.float64, device=dev) * r + r
        sL = torch.randn(K, dtype=torch.float64, device=dev) * r
        tm = timed(lambda: call('oriana_mstep_gamma_pair', ptr(q[0]), ptr(q[1]), ptr(sE), ptr(sL), float(r), ptr(q[2]), ptr(q[3]),
                                ptr(sE), ptr(sL), float(r), ptr(keep), K, st))
        print('r=%d K=%d (%.2f TB/s at 44 B per element) rpb=%s lib=%s: gamma_update_finalize %.1f us, mstep pair %.1f us' % (
            r, K, 44.0 * r * K / (t * 1e-6) / 1e12, os.environ.get('ORIANA_GU_RPB', 'auto'), os.path.basename(os.environ.get('ORIANA_HIP_LIB', 'default')), t, tm))


if __name__ == '__main__':
    main()

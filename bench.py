#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""bench.py -- CAVI sweeps/s of pCMF on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one CAVI sweep (model.step(): E-step + M-step) of pCMF (GaP) over the synthetic
1,000,000 x 30,000 count matrix with K = 100 (BASELINE.json configs[3], the configuration the
metric is quoted on; it fits one MI355X in the tiled non-zero layout).  With N > 1 the cells are
row-sharded over the ranks (strong scaling: the total problem is fixed) and the per-gene
accumulators are all-reduced over RCCL once per sweep.  The count matrix, generated on the device
from the reference generator's distribution (oriana/singlecell/generation.py:68-86) with expression
probability z = 0.10 (~90 % zeros), is resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0.  `roofline` prices the responsibility pass (row + column kernels,
timed with HIP events on their stream inside the timed region) against the ALGORITHMIC bytes of
SURVEY.md 8(d): 4 n m + 4 K (2 n + 2 m) per sweep.  `cpu_baseline` times the CPU oracle
(oracle/zq_kernels.c, 1 thread like the reference's un-parallel numba kernel) on a bounded row
sample of the same matrix (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
VALU_PEAK_TFLOPS = 157.3   # FP32 vector peak (same guide)
LDS_PEAK_GBS = 256 * 128 * 2.4   # 256 CUs x 128 B/clk x 2.4 GHz = 78.6 TB/s


def recorded_traffic(workload, world):
    """HBM bytes per launch of the pass from the committed PMC run (profiles/): the counters need
    their own rocprofv3 passes, so the bench line quotes the recorded measurement for the
    configuration it was taken on and null elsewhere."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_%s.json' % workload)
    if world != 1 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return float(json.load(f)['traffic_bytes_per_pass']['total'])
    except Exception:
        return None

WORKLOADS = {
    # name: (n_total, m, K, zero_inflation_level)
    'c4': (1000000, 30000, 100, 0.10),       # BASELINE.json configs[3] -- the metric's configuration
    'c2': (10000, 2000, 20, 0.10),           # configs[1]
    'c4_eighth': (125000, 30000, 100, 0.10),  # one rank's share of c4 at 8 GPUs
    'c4_eighth_z05': (125000, 30000, 100, 0.50),  # the same at the reference generator's default z (~53 % zeros)
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default=os.environ.get('ORIANA_BENCH_WORKLOAD', 'c4'), choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-rows', type=int, default=int(os.environ.get('ORIANA_BENCH_CPU_ROWS', '2500')))
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--chunk-rows', type=int, default=8192)
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs a torch.distributed launch (WORLD_SIZE=%d)' % (args.gpus, world))
    # ORIANA_BENCH_ONE_GPU=1 (self-test on a 1-GPU box): every rank uses cuda:0 and gloo replaces RCCL
    one_gpu = os.environ.get('ORIANA_BENCH_ONE_GPU') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    from oriana_amd import engine, dist as odist
    from oriana_amd.models import GaP
    from oriana_amd.singlecell import SyntheticCounts

    n_total, m, K, z = WORKLOADS[args.workload]
    r0, r1 = odist.shard_rows(n_total, rank, world)
    n = r1 - r0
    seed = 1234 + 1000 * 4
    t_setup = time.time()
    gen = SyntheticCounts(n_total, m, K, seed=seed, device=dev, zero_inflation_level=z, row0=r0, n=n)
    counts = engine.CountTiles.from_chunks(n, m, gen.chunk, args.chunk_rows, dev,
                                           reduce_fn=(lambda t: odist.all_reduce_sum(t)) if world > 1 else None)
    a1, b1 = gen.initial_shapes()
    model = GaP(counts, k=K, use_factors=False, init=(a1, b1), device=dev,
                process_group=(dist.group.WORLD if world > 1 else None), n_total=n_total)
    del a1, b1
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.step()
    timer = engine.KernelTimer()
    model._ws.timer = timer
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.step()
    barrier()
    elapsed = time.perf_counter() - t0
    model._ws.timer = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed

    ks = timer.summary()
    row_ms = ks.get('row_pass', (0, 0.0))[1]
    col_ms = ks.get('col_pass', (0, 0.0))[1]
    fix_ms = ks.get('fixup', (0, 0.0))[1]
    pass_ms = row_ms + col_ms + fix_ms
    nnz_total = odist.sum_int(counts.nnz, None if world == 1 else dist.group.WORLD, dev)
    # algorithmic bytes of THIS rank's launch: X read once as f32 + factor / accumulator matrices
    alg_bytes = 4.0 * n * m + 4.0 * K * (2 * n + 2 * m)
    # bytes the kernels are designed to move (tiled non-zero layout): 8 B record + 4 B s (write) in the
    # row kernel, 4 B s + 1 B row index in the column kernel, + tile pointers and factor matrices
    Kp = engine.kpad(K)
    design_bytes = counts.nnz * 17.0 + counts.nrb * counts.ncb * (2 * 257 * 4.0 + 16) + 4.0 * Kp * (3 * n + 3 * m)
    achieved = alg_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
    # the pass is bound by the CU-side rates, not by HBM: useful flops
    # (6 per non-zero and factor) and useful LDS bytes (two K-vector reads per non-zero)
    useful_tflops = 6.0 * counts.nnz * K / (pass_ms * 1e-3) / 1e12 if pass_ms > 0 else 0.0
    useful_lds_gbs = 8.0 * counts.nnz * K / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
    check = float(model.alpha1.tensor.sum().item() + model.beta1.tensor.sum().item())

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(gen, n_total, m, K, args.cpu_rows)

    if rank == 0:
        out = {
            'metric': 'CAVI sweeps/sec (pCMF, 1M x 30k, K=100)', 'value': value, 'unit': 'sweeps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%s: pCMF (GaP) CAVI sweep, %d cells x %d genes, K=%d, z=%.2f (%.1f%% zeros), '
                                   'cells row-sharded over %d GPU(s)' % (args.workload, n_total, m, K, z,
                                                                         100.0 * (1.0 - nnz_total / (float(n_total) * m)), world),
                       'n_cells': n_total, 'n_genes': m, 'K': K, 'nnz': nnz_total, 'rows_per_rank': n,
                       'parallelism': 'rows/%d' % world, 'setup_s': round(t_setup, 1),
                       'hbm_gb_rank0': round(torch.cuda.max_memory_allocated() / 1e9, 1)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': recorded_traffic(args.workload, world),
                         'kernel': 'responsibility pass = k_row_pass + k_fixup + k_col_pass (rank 0 shard)',
                         'algorithmic_bytes': alg_bytes, 'design_bytes': design_bytes,
                         'row_pass_ms': row_ms, 'col_pass_ms': col_ms, 'fixup_ms': fix_ms,
                         'achieved_design_bytes': design_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0,
                         'valu': {'achieved': useful_tflops, 'peak': VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': useful_tflops / VALU_PEAK_TFLOPS},
                         'lds': {'achieved': useful_lds_gbs, 'peak': LDS_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': useful_lds_gbs / LDS_PEAK_GBS},
                         'slot_efficiency': counts.slot_efficiency()},
            'cpu_baseline': cpu,
            'check': check,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(gen, n_total, m, K, rows):
    """Oracle (CPU restatement of gap.py:67-80, 1 thread) on the first `rows` cells of the same
    matrix; a sweep is ~100 % this loop nest in the reference, so sweeps/s is
    extrapolated linearly in n."""
    from oracle import cavi_oracle as co
    rows = min(rows, gen.n)
    X = gen.chunk(0, rows).cpu().numpy().astype(np.float32)
    rng = np.random.default_rng(0)
    lu = scipy_digamma32(rng.gamma(1.0, size=(rows, K)))
    lv = scipy_digamma32(rng.gamma(1.0, size=(m, K)))
    Zi = np.empty((rows, K), np.float32)
    Zj = np.empty((m, K), np.float32)
    t0 = time.perf_counter()
    co.zq_gap(Zi, Zj, lu, lv, np.ascontiguousarray(X))
    dt = time.perf_counter() - t0
    sweeps = 1.0 / (dt * n_total / rows)
    return {'value': sweeps, 'unit': 'sweeps/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/zq_kernels.c:zq_gap on the first %d of %d cells (%.1f s), extrapolated linearly in n; '
                      'host has %d cores, 1 used (the reference kernel is single-threaded)' % (rows, n_total, dt, os.cpu_count() or 0)}


def scipy_digamma32(a):
    import scipy.special
    return scipy.special.digamma(np.maximum(a, 1e-15).astype(np.float32)).astype(np.float32)


if __name__ == '__main__':
    main()

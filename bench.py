#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""bench.py -- CAVI sweeps/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one CAVI sweep (model.step(): E-step + M-step).  The default workload is pCMF (GaP) over the
synthetic 1,000,000 x 30,000 count matrix with K = 100 (BASELINE.json configs[3], the configuration the metric
is quoted on; it fits one MI355X in the tiled non-zero layout).  With N > 1 the cells are row-sharded over the
ranks (strong scaling: the total problem is fixed) and ONE packed all-reduce per sweep sums the per-gene
accumulators over RCCL.  `python bench.py --gpus N` without a torch.distributed launch starts the N ranks itself
(fresh child processes, started before this process touches the GPU).  The count matrix, generated on the device
from the reference generator's distribution (oriana/singlecell/generation.py:68-86) with expression probability
z = 0.10 (~90 % zeros), is resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0.  `roofline` prices the kernels of the pass (timed with HIP events on their stream
inside the timed region) against the ALGORITHMIC bytes of SURVEY.md 8(d).  `cpu_baseline` times the CPU oracle --
one full sweep of the restated reference (oracle/: the C loop nest, 1 thread like the reference's un-parallel
numba kernel, plus the NumPy updates) on a bounded row sample of the same matrix (rank 0, N = 1 only) -- and an
OpenMP all-cores variant of the loop nest, labelled as not the reference's behaviour.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# the host driver only supports dmabuf IPC: must be in the environment before the HIP runtime starts
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
VALU_PEAK_TFLOPS = 157.3   # FP32 vector peak (same guide); v_pk_fma_f32 measured: 131 TFLOP/s (tools/ubench/fma_rate.hip)
LDS_PEAK_GBS = 256 * 256 * 2.4   # 256 CUs x 256 B/clk (ds_read_b128, same guide) x 2.4 GHz = 157 TB/s
MATRIX_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak (same guide: ~2.5 PFLOP/s; never the 2:1-sparsity headline)

WORKLOADS = {
    # name: (model, n_total, m, K, zero_inflation_level, BASELINE.json config)
    'c4': ('GaP', 1000000, 30000, 100, 0.10, 'configs[3]'),          # the metric's configuration
    'c2': ('GaP', 10000, 2000, 20, 0.10, 'configs[1]'),
    'c3_zi': ('ZIGaP', 100000, 20000, 50, 0.10, 'configs[2]'),
    'c5_sparse': ('SparseGaP', 500000, 25000, 64, 0.10, 'configs[4]'),
    'c4_half': ('GaP', 500000, 30000, 100, 0.10, 'one rank\'s share of configs[3] at 2 GPUs'),
    'c4_quarter': ('GaP', 250000, 30000, 100, 0.10, 'one rank\'s share of configs[3] at 4 GPUs'),
    'c4_eighth': ('GaP', 125000, 30000, 100, 0.10, 'one rank\'s share of configs[3] at 8 GPUs'),
    'c4_eighth_z05': ('GaP', 125000, 30000, 100, 0.50, 'the same share at the reference generator\'s default z'),
    'c3_zi_z05': ('ZIGaP', 100000, 20000, 50, 0.50, 'configs[2] at the reference generator\'s default z'),
    'c5_sparse_z05': ('SparseGaP', 500000, 25000, 64, 0.50, 'configs[4] at the reference generator\'s default z'),
    # [r5] configs[2] from the reference's DEFAULT start (use_factors=True, oriana/models/base.py:15, 37-40): NMF factors as
    # initial shapes (computed on the device); the first sweeps pass through a transient with many slow-path tiles
    'c3_zi_nmf': ('ZIGaP', 100000, 20000, 50, 0.10, 'configs[2] from the reference\'s default NMF start'),
}
MODEL_LABEL = {'GaP': 'pCMF', 'ZIGaP': 'ZI-pCMF', 'SparseGaP': 'sparse pCMF'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default=os.environ.get('ORIANA_BENCH_WORKLOAD', 'c4'), choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-rows', type=int, default=int(os.environ.get('ORIANA_BENCH_CPU_ROWS', '0')),
                    help='rows of the CPU baseline sample (0: sized for ~15 s of single-thread work)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--brief', action='store_true', help='a secondary workload of the headline line: short keys, small CPU sample')
    ap.add_argument('--chunk-rows', type=int, default=8192)
    ap.add_argument('--dense-density', default=os.environ.get('ORIANA_DENSE_DENSITY', 'auto'),
                    help="pCMF: genes expressed in at least this share of the cells are evaluated on the bf16 matrix cores "
                         "(hybrid layout); 'auto' = the engine's default, 0 = sliced layout only")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` (no torchrun): start the N ranks as fresh child processes.  Nothing in this
    process has touched the GPU (torch is not even imported yet); rank r gets LOCAL_RANK r -> cuda:r."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs, errs = [], []
    # every rank's stderr goes to a NAMED file (gpurun_out/bench_ranks/rank<r>.err, or the temporary directory): if this
    # process is killed -- the driver's limit -- the ranks' diagnostics and RCCL errors survive it; replayed below otherwise
    logdir = os.path.join(ROOT, 'gpurun_out', 'bench_ranks')
    try:
        os.makedirs(logdir, exist_ok=True)
    except OSError:
        logdir = tempfile.mkdtemp(prefix='bench_ranks_')
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        ef = open(os.path.join(logdir, 'rank%d.err' % r), 'w+')
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stderr=ef))
    # poll: the first rank to fail takes its siblings down (they would otherwise block in their next collective until
    # the driver's limit); overall limit ORIANA_BENCH_TIMEOUT_S (default 3000 s).  Fresh children only are ever started
    # or killed -- this process has not touched the GPU.
    deadline = time.time() + float(os.environ.get('ORIANA_BENCH_TIMEOUT_S', '3000'))
    rc = 0
    first_bad = None
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    first_bad = procs.index(p)
        if live and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 124
            for p in live:
                p.terminate()
            t_kill = time.time() + 10
            for p in live:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if live:
            time.sleep(0.2)
    # every rank's stderr, then (on failure) one line per rank: its exit code and the last line it wrote
    tails = []
    for r, ef in enumerate(errs):
        ef.seek(0)
        text = ef.read()
        ef.close()
        if text:
            sys.stderr.write(text if text.endswith('\n') else text + '\n')
        lines = [l for l in text.splitlines() if l.strip()]
        tails.append(lines[-1][:300] if lines else '')
    if rc != 0:
        for r, p in enumerate(procs):
            code = p.poll()
            why = 'timeout' if rc == 124 and first_bad is None else ('first to fail' if r == first_bad else 'stopped after rank %s failed' % first_bad if code not in (0, None) else 'ok')
            sys.stderr.write('[bench] rank %d rc=%s (%s): %s\n' % (r, code, why, tails[r]))
    sys.exit(rc)


def algorithmic_bytes(model, n, m, K):
    """SURVEY.md 8(d), for the rows of one rank."""
    b = 4.0 * n * m + 4.0 * K * (2 * n + 2 * m)                # X once as f32; log U, log V in; Z_i, Z_j out
    if model == 'ZIGaP':
        b += 4.0 * n * m + 8.0 * n * m + 4.0 * K * m           # D_hat in the Z pass; D update (read X, write D_hat); Zlog
    if model == 'SparseGaP':
        b += 8.0 * K * m + 4.0 * K * m                         # S_tilde, S_hat; Zlog
    return b


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist

    # ORIANA_BENCH_ONE_GPU=1 (self-test on a 1-GPU box): every rank uses cuda:0 and gloo replaces RCCL
    one_gpu = os.environ.get('ORIANA_BENCH_ONE_GPU') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # ORIANA_BENCH_FORCE_PG=1 (rehearsal of the RCCL path on a 1-GPU box, tests/test_sharded_gpu.py): a process group of
    # ONE rank on the nccl backend, and ORIANA_FORCE_SHARDED=1 makes the sweep issue every collective (self all-reduces)
    force_pg = world == 1 and os.environ.get('ORIANA_BENCH_FORCE_PG') == '1'
    if force_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:                     # (a free port: two rehearsals may share a box)
            import socket
            sk = socket.socket()
            sk.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
            sk.close()
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ['ORIANA_FORCE_SHARDED'] = '1'
    if world > 1 or force_pg:
        if one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    if rank != 0:
        # ONE json line on the job's stdout: whatever a library prints on the other ranks' stdout (RCCL's version banner sits
        # in the C stdio buffer until exit) goes to their stderr
        sys.stdout.flush()
        os.dup2(2, 1)
    out = measure(args, args.workload, args.steps, args.warmup, world, rank, dev, np, torch, dist,
                  cpu_rows=(None if args.no_cpu else args.cpu_rows), brief=args.brief)
    if world > 1 or force_pg:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()                 # (rank 0: the banner leaves the buffer BEFORE the line, not at exit after it)
    if rank == 0:
        # The other single-GPU configurations of BASELINE.json ride in the SAME json line (so that the driver's run, not
        # only profiles/, carries them): configs[2] (ZI-pCMF; also from the reference's default NMF start) and configs[4]
        # (sparse pCMF), a short run each after the headline measurement, with their own roofline objects and parity slabs.
        # [r5] Each runs in a FRESH CHILD PROCESS under a wall-clock budget: a fault, an abort or a hang there cannot cost
        # the headline, which is also written to gpurun_out/bench_headline.json before they start.  ORIANA_BENCH_SECONDARY=0
        # skips them; ORIANA_BENCH_SECONDARY_BUDGET_S (default 420) bounds their total time.
        if (world == 1 and args.workload == 'c4' and not args.no_cpu and not args.brief
                and os.environ.get('ORIANA_BENCH_SECONDARY', '1') != '0'):
            try:
                os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                with open(os.path.join(ROOT, 'gpurun_out', 'bench_headline.json'), 'w') as f:
                    f.write(json.dumps(out) + '\n')
            except OSError:
                pass
            out['secondary_workloads'] = run_secondaries(('c3_zi', 'c5_sparse', 'c3_zi_nmf', 'c4_eighth_z05'))
        _flush_c_stdio()
        print(json.dumps(out))
        sys.stdout.flush()


def run_secondaries(names):
    """Each secondary workload as `python bench.py --workload <name> --brief` in a child process (this process has finished
    its GPU work and freed its memory), under what is left of a total budget; a failure becomes an {'error': ...} entry."""
    budget = float(os.environ.get('ORIANA_BENCH_SECONDARY_BUDGET_S', '420'))
    t_end = time.time() + budget
    res = []
    for wl in names:
        left = t_end - time.time()
        if left < 45:
            res.append({'workload': wl, 'error': 'skipped: %.0f s left of the secondary budget' % max(left, 0.0)})
            continue
        env = dict(os.environ, ORIANA_BENCH_SECONDARY='0')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'ORIANA_BENCH_FORCE_PG', 'ORIANA_FORCE_SHARDED'):
            env.pop(k, None)
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', wl, '--steps', '26' if wl.endswith('_nmf') else '10',
               '--warmup', '0' if wl.endswith('_nmf') else '3', '--cpu-rows', '400', '--brief']
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=left)
            lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
            if p.returncode == 0 and lines:
                res.append(json.loads(lines[-1]))
            else:
                tail = [l for l in p.stderr.splitlines() if l.strip()]
                res.append({'workload': wl, 'error': 'rc=%d: %s' % (p.returncode, (tail[-1] if tail else '')[:300])})
        except subprocess.TimeoutExpired:
            res.append({'workload': wl, 'error': 'timed out after %.0f s (child stopped)' % left})
        except Exception as exc:                # never let an extra figure break the bench line
            res.append({'workload': wl, 'error': repr(exc)[:300]})
    return res


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def measure(args, workload, steps, warmup, world, rank, dev, np, torch, dist, cpu_rows=None, brief=False):
    """One workload: setup, warm-up, the timed sweeps, the pass timings, the roofline object, the CPU baseline and the
    parity slab.  Returns the dict of the bench line (rank 0; None elsewhere).  `cpu_rows`: None = no CPU leg, 0 = sized
    for ~15 s of single-thread work, else the rows of the sample.  `brief`: a secondary workload (short keys only)."""
    from oriana_amd import engine, dist as odist
    import oriana_amd.models as models
    from oriana_amd.singlecell import SyntheticCounts

    class _A:                                   # the loop below reads args.steps / args.warmup / args.workload
        pass
    a = _A()
    a.__dict__.update(vars(args))
    a.steps, a.warmup, a.workload = steps, warmup, workload
    args = a
    mname, n_total, m, K, z, cfg_label = WORKLOADS[args.workload]
    r0, r1 = odist.shard_rows(n_total, rank, world)
    n = r1 - r0
    seed = 1234 + 1000 * 4
    torch.cuda.reset_peak_memory_stats()          # (hbm_gb_rank0 of THIS workload, not the process-wide peak)
    t_setup = time.time()
    gen = SyntheticCounts(n_total, m, K, seed=seed, device=dev, zero_inflation_level=z, row0=r0, n=n)
    dd = None
    if mname == 'GaP' and engine.dense_supported(K):
        dd = engine.auto_dense_density(n_total, m, K) if args.dense_density == 'auto' else (float(args.dense_density) or None)
    min_share = 0.0
    if mname in ('ZIGaP', 'SparseGaP') and engine.dense_supported(K):
        # the ZI / sparse models' 'auto': the dense block only when its genes hold >= 75 % of the non-zeros (models/base.py)
        if args.dense_density == 'auto':
            dd, min_share = engine.auto_dense_density(n_total, m, K), 0.75
        else:
            dd = float(args.dense_density) or None
    counts = engine.CountTiles.from_chunks(n, m, gen.chunk, args.chunk_rows, dev,
                                           reduce_fn=(lambda t: odist.all_reduce_sum(t)) if (world > 1 or odist.sharded()) else None,
                                           dense_density=dd, n_total=n_total, dense_min_share=min_share)
    nmf_start = args.workload.endswith('_nmf')
    if nmf_start:
        # the reference's DEFAULT start (use_factors=True, base.py:15, 37-40): NMF factors as initial shapes, here computed on
        # the device from the packed counts (models/deviceinit.py; the reference calls scikit-learn on the dense host matrix)
        model = getattr(models, mname)(counts, k=K, use_factors=True, init='nmf', device=dev, seed=seed,
                                       process_group=(dist.group.WORLD if (world > 1 or odist.sharded()) else None), n_total=n_total)
    else:
        a1, b1 = gen.initial_shapes()
        model = getattr(models, mname)(counts, k=K, use_factors=False, init=(a1, b1), device=dev,
                                       process_group=(dist.group.WORLD if (world > 1 or odist.sharded()) else None), n_total=n_total)
        del a1, b1
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.step()
    timer = engine.KernelTimer(prealloc=8 * args.steps)
    # The passes are timed with HIP events on the stream they run on, inside the timed region.  An event costs ~4.5 us
    # of stream time: nothing against a 50 ms sweep, a third of a 0.11 ms one -- launch-bound sizes (below 2e8 entries)
    # time the passes of every 4th sweep and take the per-sweep median from the same sweeps.
    stride = 1 if float(n_total) * m >= 2e8 else 4
    sampled = [i for i in range(args.steps) if i % stride == 0]
    marks = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in sampled}
    if world > 1 or odist.sharded():
        model._xch.timing = True         # events around every blocking exchange: allreduce_exposed_ms
    # (NMF-start workload: slow-path tiles of every sweep, summed on the device into one slot per sweep and read AFTER the timed
    #  region -- a host read per sweep would put a device-to-host round trip into every step of `value`)
    ntiles = counts.nrb * counts.ncb
    flagged_dev = torch.zeros(max(args.steps, 1), dtype=torch.int64, device=dev) if nmf_start else None
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in marks:
            model._ws.timer = timer
            marks[i][0].record()
            model.step()
            marks[i][1].record()
            model._ws.timer = None
        else:
            model.step()
        if nmf_start:
            flagged_dev[i].copy_(model._ws.tile_flag[:max(ntiles, 1)].sum())
    barrier()
    elapsed = time.perf_counter() - t0
    model._ws.timer = None
    flagged = [int(v) for v in flagged_dev.tolist()] if nmf_start else []
    exposed_ms = model._xch.exposed_ms() if (world > 1 or odist.sharded()) else None
    model._xch.timing = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    # collectives of one sweep's exchange (counted over the sweeps run so far: warm-up + timed)
    coll_per_sweep = round(model._xch.n_collectives / max(model._xch.n_reduces, 1))
    value = args.steps / elapsed
    sweep_ms = [marks[i][0].elapsed_time(marks[i][1]) for i in sorted(marks)]
    step_ms = sorted(sweep_ms)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])

    ks = {k: v[1] * v[0] / len(sampled) for k, v in timer.summary().items()}       # ms per sweep and kernel group
    pass_names = ['row_pass', 'dense_images', 'dense_row', 'fixup', 'row_spmm', 'col_pass', 'dense_col', 'col_pass_log', 'DV', 'DtU',
                  'D_update']
    pass_ms = sum(ks.get(k, 0.0) for k in pass_names)
    nnz_total = odist.sum_int(counts.nnz, None if world == 1 else dist.group.WORLD, dev)
    alg_bytes = algorithmic_bytes(mname, n, m, K)
    Kp = engine.kpad(K)
    # bytes the responsibility kernels are designed to move (tiled non-zero layout): 8 B record + 4 B s (write) in
    # the row kernel, 4 B s + 1 B row index in the column kernel, + tile pointers and factor matrices
    design_bytes = counts.nnz_sparse * 17.0 + counts.nrb * counts.ncb * (2 * 257 * 4.0 + 16) + 4.0 * Kp * (3 * n + 3 * m)
    design_bytes += 10.0 * n * counts.gd          # dense genes: 2 B count + 4 B s written, 4 B s read per entry
    achieved = alg_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
    resp_ms = sum(ks.get(k, 0.0) for k in ('row_pass', 'fixup', 'col_pass'))
    useful_tflops = 6.0 * counts.nnz_sparse * K / (resp_ms * 1e-3) / 1e12 if resp_ms > 0 else 0.0
    useful_lds_gbs = 8.0 * counts.nnz_sparse * K / (resp_ms * 1e-3) / 1e9 if resp_ms > 0 else 0.0
    check = float(model.alpha1.tensor.sum().item() + model.beta1.tensor.sum().item())
    # SURVEY 8(d)'s second regime (z = 0.5, the reference generator's default: ~53 % zeros): the pass is no longer priced by the
    # bytes of X but by its arithmetic -- the sliced non-zeros on the vector ALUs, the dense block on the bf16 matrix cores (six
    # cross products of three-way splits per float32 product: 36 K bf16 flops per dense entry for den, R and C)
    dense_ms = sum(ks.get(k, 0.0) for k in ('dense_row', 'dense_col'))
    matrix_tflops = 36.0 * K * float(n) * counts.gd / (dense_ms * 1e-3) / 1e12 if dense_ms > 0 else 0.0
    compute_bound = z >= 0.3

    # multi-GPU: per-rank pass times and the time of the sweep's packed all-reduce (measured outside the timed region)
    per_rank = None
    allreduce_ms = None
    if world > 1:
        mine = torch.tensor([ks.get('row_pass', 0.0), ks.get('col_pass', 0.0), pass_ms], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [[round(float(x), 4) for x in v.tolist()] for v in allv]
        barrier()
        ta = time.perf_counter()
        for _ in range(10):
            model._xch.reduce()
        barrier()
        allreduce_ms = (time.perf_counter() - ta) / 10 * 1e3

    # ZI workloads: the same sweeps with the three dense contractions on the float64 kernels (the arithmetic the reference's
    # np.dot uses, zigap.py:116, 124, 132) -- outside the timed region, reported beside the default (float32-equivalent) figure
    f64_ms = None
    if model.zi and getattr(model, '_fast_dense', False):
        model._fast_dense = False
        model._DV_next = None
        for _ in range(2):
            model.step()
        barrier()
        tf = time.perf_counter()
        nf = max(3, min(args.steps, 10))
        for _ in range(nf):
            model.step()
        barrier()
        f64_ms = (time.perf_counter() - tf) / nf * 1e3
        model._fast_dense = True
        model._DV_next = None

    cpu = None
    slab = None
    if rank == 0 and world == 1 and cpu_rows is not None:
        cpu, slab = cpu_baseline_and_slab(np, torch, engine, model, gen, mname, n_total, m, K, cpu_rows, dev, openmp=not brief)
    stateless = None
    if rank == 0 and world == 1 and cpu_rows is not None and mname == 'GaP' and not brief:
        stateless = stateless_binding_ms(np, torch, engine, model, gen, m, K, dev)
    twin = None
    if rank == 0 and world == 1 and cpu_rows is not None and mname in ('ZIGaP', 'SparseGaP') and not args.workload.endswith('_nmf'):
        twin = resident_twin_ms(np, torch, engine, model, gen, mname, m, K, dev)

    if rank == 0:
        traffic, traffic_src = recorded_traffic(args.workload, world, counts.gd > 0)
        frac_step = alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_per_step > 0 else 0.0
        out = {
            'metric': 'CAVI sweeps/sec (%s, %s x %s, K=%d)' % (MODEL_LABEL[mname], fmt_dim(n_total), fmt_dim(m), K),
            'value': value, 'unit': 'sweeps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'ms_per_step_median': median_ms,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%s (%s): %s (%s) CAVI sweep, %d cells x %d genes, K=%d, z=%.2f (%.1f%% zeros), '
                                   'cells row-sharded over %d GPU(s)' % (args.workload, cfg_label, MODEL_LABEL[mname], mname,
                                                                         n_total, m, K, z,
                                                                         100.0 * (1.0 - nnz_total / (float(n_total) * m)), world),
                       'n_cells': n_total, 'n_genes': m, 'K': K, 'nnz': nnz_total, 'rows_per_rank': n,
                       'parallelism': 'rows/%d' % world, 'setup_s': round(t_setup, 1),
                       'hbm_gb_rank0': round(torch.cuda.max_memory_allocated() / 1e9, 1),
                       'collectives_per_sweep': (coll_per_sweep + (1 if model.zi else 0)) if (world > 1 or odist.sharded()) else 0,
                       'layout': ('hybrid: %d genes (expressed in >= %.0f%% of the cells, %.1f%% of the non-zeros) as a dense block on the '
                                  'bf16 matrix cores (float32-equivalent: exact three-way splits, six cross products), %d genes sliced'
                                  % (counts.gd, 100.0 * counts.dense_density, 100.0 * counts.dense.nnz / max(counts.nnz, 1), counts.ms))
                                 if counts.gd else 'sliced non-zero layout'},
            'roofline': {'bound': 'valu+matrix' if compute_bound else 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         # the same algorithmic bytes over the WHOLE sweep's wall time (updates, launches and gaps included)
                         'frac_step': frac_step,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'the pass of one sweep on rank 0: ' + ' + '.join(k for k in pass_names if k in ks),
                         'algorithmic_bytes': alg_bytes, 'design_bytes': design_bytes,
                         'kernel_ms': {k: ks[k] for k in pass_names if k in ks},
                         'row_pass_ms': ks.get('row_pass', 0.0), 'col_pass_ms': ks.get('col_pass', 0.0),
                         'fixup_ms': ks.get('fixup', 0.0),
                         # the responsibility kernels are bound on the CU side: VALU issue + LDS-return
                         # traffic, not HBM -- the same launches against those rates
                         # what binds each kernel of the pass (DESIGN_HISTORY.md section 10; counters under profiles/r03_*)
                         'limiter_per_kernel': {k: v for k, v in {
                             'row_pass': 'VALU issue + LDS return port (two lanes per row, 400 B of K-vector per slot)',
                             'col_pass': 'VALU issue + LDS return port (issue-bound: masking the padding slots\' reads leaves the time unchanged, DESIGN_HISTORY.md 10 j)',
                             'dense_row': 'matrix pipe + LDS operand reads + VALU (splits, s = x / den), which add up rather than overlap',
                             'dense_col': 'matrix pipe + LDS operand reads + VALU (splits of s); HBM read of s (4 B per entry)',
                             'dense_images': 'HBM (split operand images, (n + gd) K values)', 'fixup': 'rare (exact slow path)'}.items() if k in ks},
                         'limiter': ('valu+lds (see DESIGN.md section 4)' if not model.zi else
                                     'responsibility kernels: valu+lds (DESIGN.md section 4); dense ZI kernels: valu + matrix '
                                     'cores, which barely overlap (DESIGN_HISTORY.md section 10h)'),
                         'valu': {'achieved': useful_tflops, 'peak': VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': useful_tflops / VALU_PEAK_TFLOPS},
                         'lds': {'achieved': useful_lds_gbs, 'peak': LDS_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': useful_lds_gbs / LDS_PEAK_GBS},
                         # the dense genes' kernels against the dense bf16 matrix peak (issued flops: six bf16 products per
                         # float32 product, 3 K products per entry)
                         'matrix': {'achieved': matrix_tflops, 'peak': MATRIX_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s (bf16, issued)',
                                    'frac': matrix_tflops / MATRIX_BF16_PEAK_TFLOPS,
                                    'dense_entries': int(n) * int(counts.gd), 'dense_ms': dense_ms},
                         'slot_efficiency': counts.slot_efficiency()},
            'cpu_baseline': cpu,
            'parity_slab': slab,
            'f64_reference_arithmetic_ms': f64_ms,
            'check': check,
        }
        # launch-bound sizes: the same sweep replayed from a captured hipGraph (after, and outside, the timed region; the
        # headline `value` stays the eager figure).  Informative only: at configs[1] a sweep is ~30 launches of a few
        # microseconds of work each.
        out['pass_timing'] = ('HIP events around every pass of every sweep of the timed region' if stride == 1 else
                              'HIP events around the passes of every %dth sweep of the timed region (%d of %d sweeps)'
                              % (stride, len(sampled), args.steps))
        if world == 1 and float(n_total) * m < 2e8:
            # the same eager sweeps without any event record (what a user's fit() loop runs), after the timed region
            torch.cuda.synchronize()
            tu = time.perf_counter()
            reps = max(50, args.steps)
            for _ in range(reps):
                model.step()
            torch.cuda.synchronize()
            out['uninstrumented_ms_per_step'] = (time.perf_counter() - tu) / reps * 1e3
        if world == 1 and not model.zi and float(n_total) * m <= 1e8:
            try:
                model._ws.timer = None
                model.capture_graph()
                for _ in range(5):
                    model.step()
                torch.cuda.synchronize()
                tg = time.perf_counter()
                reps = max(20, args.steps)
                for _ in range(reps):
                    model.step()
                torch.cuda.synchronize()
                out['graph_replay_ms_per_step'] = (time.perf_counter() - tg) / reps * 1e3
            except Exception as exc:                 # never let the extra figure break the bench line
                out['graph_replay_ms_per_step'] = None
                out['graph_replay_error'] = repr(exc)[:200]
        if world > 1:
            out['per_rank_ms'] = {'columns': ['row_pass', 'col_pass', 'pass'], 'ranks': per_rank}
            out['allreduce_ms'] = allreduce_ms
            # rank 0's mean wait at the exchange inside the timed sweeps (compute stream reaches the exchange -> last collective
            # done): what is NOT hidden under the column pass / the dense gene-side kernel
            out['allreduce_exposed_ms'] = exposed_ms
            out['allreduce_share_of_step'] = allreduce_ms / ms_per_step if ms_per_step > 0 else None
            out['exchange_bytes'] = int(model._xch.numel * 4)
            out['exchange'] = ('one step per sweep: float64 all-reduce of the (small) rate partials, started before the column pass; float32 '
                               'all-reduce of the per-gene sums -- pCMF with K == Kp: in packed gene order, the sliced genes\' segment started '
                               'before the dense gene-side kernel, the dense genes\' segment after it')
        if stateless is not None:
            out['stateless_binding'] = stateless
        if twin is not None:
            out['resident_binding'] = twin
        if nmf_start:
            settled = sorted(sweep_ms[-5:])[len(sweep_ms[-5:]) // 2]
            out['transient'] = {'what': 'sweeps 0..%d after the reference\'s default start (use_factors=True: NMF factors as initial '
                                        'shapes, base.py:37-40): ms of every sweep (HIP events), tiles with slow-path entries' % (args.steps - 1),
                                'sweep_ms': [round(v, 3) for v in sweep_ms], 'mean_ms': sum(sweep_ms) / len(sweep_ms),
                                'max_ms': max(sweep_ms), 'max_ms_from_sweep_1': max(sweep_ms[1:]) if len(sweep_ms) > 1 else None,
                                'settled_ms': settled, 'flagged_tiles': flagged, 'tiles': int(ntiles)}
        if world == 1 and odist.sharded():
            out['exchange_rehearsal'] = {'backend': dist.get_backend(), 'ranks': 1, 'exchanges': int(model._xch.n_reduces),
                                         'exposed_ms': exposed_ms,
                                         'collectives': int(model._xch.n_collectives),
                                         'what': 'one-rank process group: every collective of the sharded sweep issued as a self all-reduce'}
        if brief:
            keep = ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median', 'dtype', 'config',
                    'roofline', 'parity_slab', 'f64_reference_arithmetic_ms', 'cpu_baseline', 'check', 'transient', 'resident_binding')
            out = {k: out[k] for k in keep if k in out}
            out['workload'] = workload
    else:
        out = None
    # free the workload before the next one is set up (the headline matrix holds ~65 GB)
    del model, counts, gen, timer
    torch.cuda.empty_cache()
    return out


def fmt_dim(v):
    if v % 1000000 == 0:
        return '%dM' % (v // 1000000)
    if v % 1000 == 0:
        return '%dk' % (v // 1000)
    return str(v)


def recorded_traffic(workload, world, hybrid=False):
    """HBM bytes per launch of the pass from the committed PMC run (profiles/): the counters need their own rocprofv3
    passes, so the bench line quotes the recorded measurement for the configuration AND LAYOUT it was taken on (with its
    source) and null elsewhere."""
    if world != 1:
        return None, None
    # (the NMF start runs the kernels of configs[2] on the same matrix: its settled sweeps move what the c3_zi counter run counted)
    same_as = {'c3_zi_nmf': 'c3_zi'}
    tag = same_as.get(workload, workload) + ('_hybrid' if hybrid else '')
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        path = os.path.join(ROOT, 'profiles', '%s_pmc_hbm_%s.json' % (rnd, tag))
        if os.path.exists(path):
            try:
                with open(path) as f:
                    src = 'recorded: profiles/' + os.path.basename(path)
                    if workload in same_as:
                        src += ' (the counter run of %s: same matrix, same kernels)' % same_as[workload]
                    return float(json.load(f)['traffic_bytes_per_pass']['total']), src
            except Exception:
                return None, None
    return None, None


def stateless_binding_ms(np, torch, engine, model, gen, m, K, dev, rows=2048):
    """The drop-in kernel boundary as INTEGRATION.md section B binds it.  (1) oriana_zq_gap_f32: dense float32 device
    matrices in the reference's argument order, no resident state -- every call packs X into the sliced layout, runs the
    pass and synchronises.  (2) [r5] the RESIDENT handle: oriana_counts_create_dense_f32 once (pack + plans in C host code, the
    layout of the run: hybrid when the model's is), then oriana_zq_gap_resident per call.  (3) the same slab through the
    Python host path the models use (engine.CountTiles + engine.zq_gap), timed the same way.  ms per call on the first
    `rows` cells."""
    import ctypes
    rows = min(rows, gen.n)
    X = gen.chunk(0, rows).to(torch.float32).contiguous()
    lu = model._log_U_hat[:rows].contiguous()
    lv = model._log_V_hat.contiguous()
    Zi = torch.empty(rows, K, device=dev); Zj = torch.empty(m, K, device=dev)

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    ms = timed(lambda: engine.zq_gap_stateless(Zi, Zj, lu, lv, X), 5)
    out = {'entry': 'oriana_zq_gap_f32 (csrc/stateless.hip)', 'rows': rows, 'genes': m, 'K': K, 'ms_per_call': ms,
           'dense_x_gb_per_s': 4.0 * rows * m / (ms * 1e-3) / 1e9,
           'note': 'packs the dense X on every call (sliced layout only, no matrix-core path) and synchronises; the resident '
                   'handle / the resident model path amortise the packing over the sweeps'}
    try:
        lib = engine._lib.load()
        dd = float(getattr(model.counts, 'dense_density', None) or 0.0)
        h = ctypes.c_void_p(None)
        t0 = time.perf_counter()
        engine.call('oriana_counts_create_dense_f32', ctypes.addressof(h), engine.ptr(X), rows, m, m, K, dd, engine.stream_ptr())
        torch.cuda.synchronize()
        create_ms = (time.perf_counter() - t0) * 1e3
        st = engine.stream_ptr()
        res_ms = timed(lambda: engine.call('oriana_zq_gap_resident', h, engine.ptr(Zi), engine.ptr(Zj), engine.ptr(lu), engine.ptr(lv), st), 20)
        Zi_r, Zj_r = Zi.clone(), Zj.clone()
        info = (ctypes.c_int64 * 13)()
        lib.oriana_counts_info(h, info, 13)
        lib.oriana_counts_destroy(h)
        ct = engine.CountTiles.from_dense(X, dev, dense_density=dd or None)
        ws = engine.ZWorkspace(ct, K)
        py_ms = timed(lambda: engine.zq_gap(ws, Zi, Zj, lu, lv), 20)
        torch.cuda.synchronize()
        scale = float(Zj.abs().max().item()) or 1.0
        out.update({'resident_entry': 'oriana_counts_create_dense_f32 + oriana_zq_gap_resident (csrc/resident.hip)',
                    'resident_create_ms': create_ms, 'resident_ms_per_call': res_ms, 'model_path_ms_per_call': py_ms,
                    'resident_vs_model_path': res_ms / py_ms if py_ms > 0 else None,
                    'resident_dense_genes': int(info[5]), 'resident_bytes': int(info[8]),
                    'resident_vs_model_path_max_abs_diff_rel': float(max((Zi_r - Zi).abs().max().item(), (Zj_r - Zj).abs().max().item()) / scale)})
    except Exception as exc:                        # never let the extra figure break the bench line
        out['resident_error'] = repr(exc)[:300]
    return out


def resident_twin_ms(np, torch, engine, model, gen, mname, m, K, dev, rows=2048):
    """[r6] The resident C-ABI handle for the nests of the ZI / sparse models: oriana_counts_create_dense_f32
    once + oriana_counts_declare_unit_dropout, then oriana_zq_zigap_resident (zigap.py:105-112; third output NULL, as the model
    classes skip it) / oriana_zq_sparse_gap_resident (sparse_gap.py:107-115) per call, beside the same slab through the Python
    host path of the models (engine.CountTiles + engine.zq), timed the same way.  ms per call on the first `rows` cells."""
    import ctypes
    rows = min(rows, gen.n)
    X = gen.chunk(0, rows).to(torch.float32).contiguous()
    lu = model._log_U_hat[:rows].contiguous()
    lv = model._log_V_hat.contiguous()
    Zi = torch.empty(rows, K, device=dev); Zj = torch.empty(m, K, device=dev); Zl = torch.empty(m, K, device=dev)

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    out = {'rows': rows, 'genes': m, 'K': K}
    try:
        lib = engine._lib.load()
        dd = float(getattr(model.counts, 'dense_density', None) or 0.0) if model.counts.gd else 0.0
        h = ctypes.c_void_p(None)
        engine.call('oriana_counts_create_dense_f32', ctypes.addressof(h), engine.ptr(X), rows, m, m, K, dd, engine.stream_ptr())
        engine.call('oriana_counts_declare_unit_dropout', h, 1)
        st = engine.stream_ptr()
        ct = engine.CountTiles.from_dense(X, dev, dense_density=dd or None)
        ws = engine.ZWorkspace(ct, K)
        if mname == 'ZIGaP':
            D = model._Dp[:rows, :m].contiguous()
            quirk = 1 if model.reference_quirks else 0
            out['entry'] = 'oriana_zq_zigap_resident (csrc/resident.hip), D_hat == 1 at the non-zero counts declared, third output NULL'
            res = lambda: engine.call('oriana_zq_zigap_resident', h, engine.ptr(Zi), engine.ptr(Zj), None, engine.ptr(lu), engine.ptr(lv),
                                      engine.ptr(D), quirk, st)
            dq = torch.empty(rows, K, dtype=torch.float32, device=dev) if quirk else None

            def py():
                if quirk:
                    engine.call('oriana_take_cols_f32', engine.ptr(dq), engine.ptr(D), rows, m, K, engine.stream_ptr())
                engine.zq(ws, Zi, Zj, None, lu, lv, dq=dq)
        else:
            model._threshold()
            St, Sh = model._S_tilde.contiguous(), model._S_hat.contiguous()
            out['entry'] = 'oriana_zq_sparse_gap_resident (csrc/resident.hip)'
            res = lambda: engine.call('oriana_zq_sparse_gap_resident', h, engine.ptr(Zi), engine.ptr(Zj), engine.ptr(Zl), engine.ptr(lu),
                                      engine.ptr(lv), engine.ptr(St), engine.ptr(Sh), st)
            py = lambda: engine.zq(ws, Zi, Zj, Zl, lu, lv, S_tilde=St, S_hat=Sh)
        res_ms = timed(res, 20)
        r_out = (Zi.clone(), Zj.clone(), Zl.clone())
        py_ms = timed(py, 20)
        torch.cuda.synchronize()
        lib.oriana_counts_destroy(h)
        scale = float(Zj.abs().max().item()) or 1.0
        diff = max((r_out[0] - Zi).abs().max().item(), (r_out[1] - Zj).abs().max().item())
        if mname != 'ZIGaP':
            diff = max(diff, (r_out[2] - Zl).abs().max().item() / (float(Zl.abs().max().item()) or 1.0) * scale)
        out.update({'resident_ms_per_call': res_ms, 'model_path_ms_per_call': py_ms,
                    'resident_vs_model_path': res_ms / py_ms if py_ms > 0 else None,
                    'resident_vs_model_path_max_abs_diff_rel': float(diff / scale)})
    except Exception as exc:                        # never let the extra figure break the bench line
        out['resident_error'] = repr(exc)[:300]
    return out


def cpu_baseline_and_slab(np, torch, engine, model, gen, mname, n_total, m, K, rows, dev, openmp=True):
    """(1) CPU baseline: one FULL sweep of the oracle (oracle/cavi_oracle.py: the C restatement of the loop nest,
    1 thread, + the NumPy/SciPy updates) on the first `rows` cells of the same matrix, extrapolated to n_total
    cells (loop nest linear in the rows; updates linear in rows + genes); BASELINE configs[1] runs in full.
    Also the loop nest with OpenMP on every host core (NOT the reference's behaviour: its kernel is one thread).
    (2) Parity on a slab of the workload itself: Z_i of those rows and their contribution to Z_j from the MODEL's
    current E[log U], E[log V], HIP against the oracle."""
    from oracle import cavi_oracle as co
    import scipy.special
    cls = {'GaP': co.OracleGaP, 'ZIGaP': co.OracleZIGaP, 'SparseGaP': co.OracleSparseGaP}[mname]
    full = (n_total * m <= 4e7)
    if rows <= 0:
        rows = n_total if full else max(64, int(15.0 / (m * K * 2.3e-9)))      # ~2.3 ns per (i, j, k): ~15 s of loop nest
    rows = min(rows, gen.n, n_total)
    X = gen.chunk(0, rows).cpu().numpy()
    rng = np.random.default_rng(0)
    a1 = rng.gamma(1.0, size=(rows, K))
    b1 = rng.gamma(1.0, size=(m, K))
    ref = cls(X.astype(np.int64), K, a1, b1)
    t0 = time.perf_counter()
    ref.step()
    t_step = time.perf_counter() - t0
    # the loop nest alone on the same arrays (what the reference spends ~100 % of a sweep in)
    lu = scipy.special.digamma(np.maximum(a1, 1e-15).astype(np.float32)).astype(np.float32)
    lv = scipy.special.digamma(np.maximum(b1, 1e-15).astype(np.float32)).astype(np.float32)
    Xf = np.ascontiguousarray(X.astype(np.float32))
    Zi = np.empty((rows, K), np.float32)
    Zj = np.empty((m, K), np.float32)
    t0 = time.perf_counter()
    co.zq_gap(Zi, Zj, lu, lv, Xf)
    t_loop = time.perf_counter() - t0
    t_upd = max(t_step - t_loop, 0.0) if mname == 'GaP' else 0.0
    t_nest = t_step - t_upd
    sweep_s = t_nest * (n_total / rows) + t_upd * ((n_total + m) / float(rows + m))
    cores = os.cpu_count() or 1
    omp = None
    try:
        if not openmp:
            raise RuntimeError('skipped for a secondary workload')
        # its own, larger sample: 2000 rows would leave 8 rows per thread on a 256-core host
        rows_omp = min(gen.n, n_total, max(rows, min(16384, rows * max(1, cores // 8))))
        Xo = Xf if rows_omp == rows else np.ascontiguousarray(gen.chunk(0, rows_omp).cpu().numpy().astype(np.float32))
        luo = lu if rows_omp == rows else scipy.special.digamma(np.maximum(rng.gamma(1.0, size=(rows_omp, K)), 1e-15).astype(np.float32)).astype(np.float32)
        Zio = np.empty((rows_omp, K), np.float32)
        t0 = time.perf_counter()
        co.zq_gap_omp(Zio, Zj, luo, lv, Xo, cores)
        t_omp = time.perf_counter() - t0
        omp = {'value': 1.0 / (t_omp * n_total / rows_omp + t_upd * ((n_total + m) / float(rows + m))), 'unit': 'sweeps/s',
               'cores': cores, 'note': 'NOT the reference\'s behaviour (its numba kernel is single-threaded, gap.py:67): the '
                                       'pCMF loop nest with OpenMP over the cells, %.2f s on %d cells, extrapolated linearly' % (t_omp, rows_omp)}
        del Xo, Zio
    except Exception as e:                                                     # no OpenMP runtime on this host
        omp = {'error': repr(e)}
    cpu = {'value': 1.0 / sweep_s, 'unit': 'sweeps/s', 'cores': 1, 'kind': 'port',
           'sample': 'oracle sweep (%s.step(): oracle/zq_kernels.c loop nest + NumPy updates) on %s %d of %d cells: '
                     '%.1f s (loop nest %.1f s), extrapolated linearly; host has %d cores, 1 used (the reference kernel is '
                     'single-threaded)' % (cls.__name__, 'ALL' if rows == n_total else 'the first', rows, n_total, t_step, t_nest, cores),
           'openmp_all_cores': omp}
    # ---- parity slab: the WORKLOAD's own loop nest at its K, with the model's current factors (and D_hat rows / masks) ----
    slab = None
    try:
        srows = min(rows, 2500, gen.n)
        Xs = np.ascontiguousarray(X[:srows].astype(np.float32))
        lus = np.ascontiguousarray(model._log_U_hat[:srows].cpu().numpy())
        lvs = np.ascontiguousarray(model._log_V_hat.cpu().numpy())
        Zi_o = np.empty((srows, K), np.float32)
        Zj_o = np.empty((m, K), np.float32)
        Zl_o = np.empty((m, K), np.float32)
        # (the slab goes through the layout of the run: with a hybrid layout its densest genes take the matrix-core path)
        ct = engine.CountTiles.from_dense(torch.from_numpy(Xs).to(dev), dev, dense_density=getattr(model.counts, 'dense_density', None))
        ws = engine.ZWorkspace(ct, K)
        Zi_h = torch.empty(srows, K, device=dev)
        Zj_h = torch.empty(m, K, device=dev)
        Zl_h = None
        lus_d, lvs_d = torch.from_numpy(lus).to(dev), torch.from_numpy(lvs).to(dev)
        extra = {}
        if mname == 'GaP':
            nest = 'gap.py:67-80'
            co.zq_gap(Zi_o, Zj_o, lus, lvs, Xs)
            engine.zq_gap(ws, Zi_h, Zj_h, lus_d, lvs_d)
        elif mname == 'ZIGaP':
            nest = 'zigap.py:79-95 (D_hat rows of the model, the D_hat[i, k] index of zigap.py:94 as the model runs it)'
            Dh = model._D_hat[:srows].contiguous()
            Dh_host = Dh.cpu().numpy()
            co.zq_zigap(Zi_o, Zj_o, Zl_o, lus, lvs, Dh_host, Xs, quirk=bool(model.reference_quirks))
            dq = Dh[:, :K].contiguous() if model.reference_quirks else None
            engine.zq(ws, Zi_h, Zj_h, None, lus_d, lvs_d, dq=dq)
            # the slab's rate terms of the sweep against float64 NumPy (zigap.py:116, 124, 131-136): a2 - alpha2 = D_hat V_hat,
            # the slab's contribution to D_hat^T U_hat, and the rows of p_d (row-local given V_hat, pi_d)
            U64 = model._U_hat[:srows].contiguous()
            V64 = model._V_hat.contiguous()
            Uh, Vh = U64.cpu().numpy(), V64.cpu().numpy()
            DV_ref = Dh_host.astype(np.float64) @ Vh
            DtU_ref = Dh_host.astype(np.float64).T @ Uh
            DtU_h = torch.zeros(m, K, dtype=torch.float64, device=dev)
            if model._fast_dense:
                scr = torch.zeros(int(engine._lib.load().oriana_dense_t_scratch_floats(srows, K)), dtype=torch.float32, device=dev)
                engine.call('oriana_dense_t_times_factor_f32', engine.ptr(DtU_h), engine.ptr(Dh), engine.ptr(U64), engine.ptr(scr),
                            model._matrix_arith, srows, m, K, engine.stream_ptr())
            else:
                engine.call('oriana_dense_times_factor', engine.ptr(DtU_h), engine.ptr(Dh), engine.ptr(U64), srows, m, K, 1, engine.stream_ptr())
            DV_h = torch.zeros(srows, K, dtype=torch.float64, device=dev)
            engine.call('oriana_dense_times_factor', engine.ptr(DV_h), engine.ptr(Dh), engine.ptr(V64), srows, m, K, 0, engine.stream_ptr())
            pi = model.pi_d.tensor.cpu().numpy()
            with np.errstate(all='ignore'):
                p_ref = co.sigmoid(co.logit(pi)[None, :] - Uh @ Vh.T)
            p_ref[:, pi <= 0] = 1e-10
            p_ref[:, pi >= 1] = 1. - 1e-10
            p_ref[Xs != 0] = 1. - 1e-10
            D_new = torch.empty(srows, m, dtype=torch.float32, device=dev)
            nzm = torch.zeros(((srows + 31) // 32) * m, dtype=torch.int32, device=dev)
            engine.call('oriana_nzmask_f32', engine.ptr(nzm), engine.ptr(torch.from_numpy(Xs).to(dev)), srows, m, engine.stream_ptr())
            psum = torch.zeros(m, dtype=torch.float64, device=dev)
            if model._fast_dense:
                lg = torch.zeros(int(engine._lib.load().oriana_dropout_sweep_scratch_floats(m, K)), dtype=torch.float32, device=dev)
                # (with V_next and the per-lane flags: the kernel the sweep itself launches, csrc/dense_zi.hip for K = 33 .. 100)
                DV_new = torch.zeros(srows, K, dtype=torch.float64, device=dev)
                engine.dropout_sweep(D_new, U64, V64, model.pi_d.tensor, nzm, engine.nzmask_tiles(nzm, srows, m), psum, V64, DV_new, lg,
                                     model._matrix_arith, srows, m, K)
            else:
                engine.call('oriana_dropout_update_fused', None, engine.ptr(D_new), engine.ptr(U64), engine.ptr(V64),
                            engine.ptr(model.pi_d.tensor), engine.ptr(nzm), engine.ptr(psum), srows, m, K, engine.stream_ptr())
            torch.cuda.synchronize()
            extra = {'DV_rel': float(np.abs(DV_h.cpu().numpy() - DV_ref).max() / max(np.abs(DV_ref).max(), 1e-300)),
                     'DtU_rel': float(np.abs(DtU_h.cpu().numpy() - DtU_ref).max() / max(np.abs(DtU_ref).max(), 1e-300)),
                     'D_hat_abs': float(np.abs(D_new.cpu().numpy().astype(np.float64) - p_ref.astype(np.float32)).max()),
                     'p_d_colsum_rel': float(np.abs(psum.cpu().numpy() - p_ref.sum(0)).max() / max(p_ref.sum(0).max(), 1e-300)),
                     'dense_arithmetic': 'float64' if not model._fast_dense else ('bf16x3' if model._matrix_arith == 1 and K <= 100 else 'f32 matrix instruction')}
        else:
            nest = 'sparse_gap.py:81-97 (S_tilde, S_hat of the model)'
            model._threshold()                       # S_tilde = p_s > tau as the NEXT sweep will take it (sparse_gap.py:113; idempotent)
            St = np.ascontiguousarray(model._S_tilde.cpu().numpy())
            Sh = np.ascontiguousarray(model._S_hat.cpu().numpy())
            co.zq_sparse_gap(Zi_o, Zj_o, Zl_o, lus, lvs, St, Sh, Xs)
            Zl_h = torch.empty(m, K, device=dev)
            engine.zq(ws, Zi_h, Zj_h, Zl_h, lus_d, lvs_d, S_tilde=model._S_tilde, S_hat=model._S_hat)
        torch.cuda.synchronize()

        def colrel(got, ref):
            cm = np.abs(ref).max(axis=0, keepdims=True)
            return float((np.abs(got.astype(np.float64) - ref) / (np.abs(ref) + cm + 1e-300)).max())
        def strictrel(got, ref):
            # strictly element-wise |d| / |ref| over the entries that are not negligible in their column (|ref| >= 1e-6 of
            # the column's largest: below that an entry is rounding noise of the sums it belongs to)
            cm = np.abs(ref).max(axis=0, keepdims=True)
            big = np.abs(ref) >= 1e-6 * cm + 1e-300
            if not big.any():
                return 0.0
            return float((np.abs(got.astype(np.float64) - ref)[big] / np.abs(ref)[big]).max())
        slab = {'rows': srows, 'what': 'the workload\'s loop nest (%s) on the first rows, from the model\'s current E[log U], E[log V]: '
                                       'HIP vs oracle, max |d| / (|ref| + colmax|ref|); *_strict: max |d| / |ref| element-wise over the '
                                       'entries with |ref| >= 1e-6 of their column\'s largest' % nest,
                'Z_i': colrel(Zi_h.cpu().numpy(), Zi_o.astype(np.float64)),
                'Z_j': colrel(Zj_h.cpu().numpy(), Zj_o.astype(np.float64)),
                'Z_i_strict': strictrel(Zi_h.cpu().numpy(), Zi_o.astype(np.float64)),
                'Z_j_strict': strictrel(Zj_h.cpu().numpy(), Zj_o.astype(np.float64)),
                'dense_genes': int(ct.gd)}
        if mname == 'GaP':
            slab['conservation'] = float(abs(float(Zi_h.double().sum().item()) - float(Xs.astype(np.float64).sum())) / max(float(Xs.sum()), 1.0))
        if Zl_h is not None:
            slab['Z_log'] = colrel(Zl_h.cpu().numpy(), Zl_o.astype(np.float64))
            slab['Z_log_strict'] = strictrel(Zl_h.cpu().numpy(), Zl_o.astype(np.float64))
        slab.update(extra)
        # ---- [r5] one SWEEP on the slab's cells: the model takes its next step() from the very state the oracle nest above was
        # evaluated on, and the cell-side update of those rows (gap.py:96-102 / zigap.py:114-120 / sparse_gap.py:117-124:
        # a1 = alpha1 + Z_i, a2 = alpha2 + rate, then Gamma.mean / Gamma.meanlog) is restated from the ORACLE's Z_i
        try:
            al1 = model.alpha1.tensor.cpu().numpy().copy(); al2 = model.alpha2.tensor.cpu().numpy().copy()
            if mname == 'GaP':
                rate = model._V_hat.sum(0).cpu().numpy()[None, :]
            elif mname == 'SparseGaP':
                rate = (model._S_hat.double() * model._V_hat).sum(0).cpu().numpy()[None, :]
            else:
                rate = DV_ref
            model.step()
            torch.cuda.synchronize()
            a1_ref = co.clamp(al1[None, :] + Zi_o)
            a2_ref = co.clamp(np.broadcast_to(al2[None, :] + rate, a1_ref.shape).copy())
            got = {'a1': model.a1.tensor[:srows].cpu().numpy(), 'a2': model.a2.tensor[:srows].cpu().numpy(),
                   'U_hat': model._U_hat[:srows].cpu().numpy(), 'log_U_hat': model._log_U_hat[:srows].cpu().numpy()}
            ref = {'a1': a1_ref, 'a2': a2_ref, 'U_hat': co.gamma_mean(a1_ref, a2_ref), 'log_U_hat': co.gamma_meanlog(a1_ref, a2_ref)}
            slab['sweep_cell_side'] = {k: colrel(got[k], np.asarray(ref[k], dtype=np.float64)) for k in got}
            slab['sweep_cell_side']['what'] = ('model.step() from that state: a1, a2, U_hat, log_U_hat of the slab\'s cells after the sweep '
                                               'against the update restated from the oracle\'s Z_i (column metric)')
        except Exception as e2:
            slab['sweep_cell_side'] = {'error': repr(e2)[:200]}
    except Exception as e:
        import traceback
        slab = {'error': repr(e), 'trace': traceback.format_exc()[-600:]}
    return cpu, slab


if __name__ == '__main__':
    main()

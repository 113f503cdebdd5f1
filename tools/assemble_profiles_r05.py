# -*- coding: utf-8 -*-
"""Files the results of tools/evidence_r05.sh (gpurun_out/r05/, the end-of-round state of the code) under profiles/r05_*."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'r05') + '/'
P = os.path.join(ROOT, 'profiles') + '/'

cp = {'bench_c4.json': 'r05_bench_c4.json', 'bench_c4_sliced.json': 'r05_bench_c4_sliced.json', 'bench_c2.json': 'r05_bench_c2.json',
      'bench_c3_zi.json': 'r05_bench_c3_zi.json', 'bench_c5_sparse.json': 'r05_bench_c5_sparse.json',
      'bench_c3_zi_nmf.json': 'r05_bench_c3_zi_nmf.json',
      'bench_c4_half.json': 'r05_bench_c4_half.json', 'bench_c4_quarter.json': 'r05_bench_c4_quarter.json',
      'bench_c4_eighth.json': 'r05_bench_c4_eighth.json', 'bench_c4_eighth_rccl1.json': 'r05_bench_c4_eighth_rccl_one_rank.json',
      'bench_c4_unfused_prep.json': 'r05_bench_c4_unfused_prep.json', 'bench_c4_r4_updates.json': 'r05_bench_c4_r4_updates.json',
      'scaling_projection.json': 'r05_scaling_projection.json', 'scaling_projection.md': 'r05_scaling_projection.md',
      'kernel_stats_c4.csv': 'r05_bench_c4_kernel_stats.csv', 'kernel_stats_c3_zi.csv': 'r05_zigap_c3_kernel_stats.csv',
      'kernel_stats_c5_sparse.csv': 'r05_sparsegap_c5_kernel_stats.csv', 'parity_errors.json': 'r05_parity_errors.json',
      'zi_trace_nmf.txt': 'r05_zigap_slow_path_trace_nmf.txt', 'sparse_k100.txt': 'r05_sparsegap_k100.txt',
      'perf_gamma.txt': 'r05_perf_gamma.txt', 'perf_gamma_r4kernel.txt': 'r05_perf_gamma_r4kernel.txt',
      'r05_c4_updates_fetch.json': 'r05_pmc_c4_updates_fetch.json', 'r05_c4_updates_write.json': 'r05_pmc_c4_updates_write.json'}
for a, b in cp.items():
    if os.path.exists(F + a):
        text = open(F + a, 'rb').read()
        if a.endswith('.json') and a.startswith('bench_'):
            lines = [l for l in text.strip().splitlines() if l.startswith(b'{')]
            if not lines:
                continue
            text = lines[-1] + b'\n'
        if a.endswith('.txt'):
            text = b'\n'.join(l for l in text.splitlines() if b'amdgpu.ids' not in l) + b'\n'
        open(P + b, 'wb').write(text)
        print(b)

# -*- coding: utf-8 -*-
"""Files the results of tools/evidence_r03b.sh (gpurun_out/r03b/, the end-of-round state of the code) under profiles/r03_*."""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'r03b') + '/'
P = os.path.join(ROOT, 'profiles') + '/'

cp = {'bench_c4.json': 'r03_bench_c4.json', 'bench_c2.json': 'r03_bench_c2.json', 'bench_c3_zi.json': 'r03_bench_c3_zi.json',
      'bench_c5_sparse.json': 'r03_bench_c5_sparse.json', 'bench_c4_eighth.json': 'r03_bench_c4_eighth.json',
      'kernel_stats_c4.csv': 'r03_bench_c4_kernel_stats.csv', 'kernel_stats_c3_zi.csv': 'r03_zigap_c3_kernel_stats.csv',
      'kernel_stats_c2.csv': 'r03_bench_c2_kernel_stats.csv', 'c2_sweep_timeline.txt': 'r03_c2_sweep_timeline.txt',
      'c2_host_vs_gpu.txt': 'r03_c2_host_vs_gpu.txt', 'perf_small.txt': 'r03_c2_passes_by_split.txt',
      'perf_small_r2kernels.txt': 'r03_c2_passes_by_split_four_lane_kernels.txt', 'zi_dense_per_k.txt': 'r03_zi_dense_per_k.txt',
      'zi_dense_per_k_round2_kernels.txt': 'r03_zi_dense_per_k_round2_kernels.txt',
      'zigap_c4shape_k100.txt': 'r03_zigap_c4shape_k100.txt', 'parity_errors.json': 'r03_parity_errors.json'}
for a, b in cp.items():
    if os.path.exists(F + a):
        text = open(F + a, 'rb').read()
        if a.endswith('.json') and a.startswith('bench_'):
            text = text.strip().splitlines()[-1] + b'\n'
        if a.endswith('.txt'):
            text = b'\n'.join(l for l in text.splitlines() if b'amdgpu.ids' not in l) + b'\n'
        open(P + b, 'wb').write(text)
        print(b)

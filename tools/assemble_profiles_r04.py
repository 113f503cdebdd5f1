# -*- coding: utf-8 -*-
"""Files the results of tools/evidence_r04.sh (gpurun_out/r04/, the end-of-round state of the code) under profiles/r04_*."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'r04') + '/'
P = os.path.join(ROOT, 'profiles') + '/'

cp = {'bench_c4.json': 'r04_bench_c4.json', 'bench_c4_sliced.json': 'r04_bench_c4_sliced.json', 'bench_c2.json': 'r04_bench_c2.json',
      'bench_c3_zi.json': 'r04_bench_c3_zi.json', 'bench_c5_sparse.json': 'r04_bench_c5_sparse.json',
      'bench_c4_eighth.json': 'r04_bench_c4_eighth.json', 'bench_c3_zi_r3kernels.json': 'r04_bench_c3_zi_r3kernels.json',
      'bench_c5_sparse_r3kernels.json': 'r04_bench_c5_sparse_r3kernels.json',
      'bench_c4_eighth_rccl1.json': 'r04_bench_c4_eighth_rccl_one_rank.json',
      'kernel_stats_c4.csv': 'r04_bench_c4_kernel_stats.csv', 'kernel_stats_c3_zi.csv': 'r04_zigap_c3_kernel_stats.csv',
      'kernel_stats_c5_sparse.csv': 'r04_sparsegap_c5_kernel_stats.csv', 'zigap_gene_count.txt': 'r04_zigap_gene_count.txt',
      'parity_errors.json': 'r04_parity_errors.json', 'bench_c4_norounds.json': 'r04_bench_c4_norounds.json',
      'bench_c3_zi_norounds.json': 'r04_bench_c3_zi_norounds.json', 'bench_c5_sparse_norounds.json': 'r04_bench_c5_sparse_norounds.json',
      'bench_c3_zi_fixedden.json': 'r04_bench_c3_zi_fixedden.json', 'bench_c3_zi_60.json': 'r04_bench_c3_zi_60.json',
      'bench_c3_zi_60_fixedden.json': 'r04_bench_c3_zi_60_fixedden.json'}
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

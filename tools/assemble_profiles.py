# -*- coding: utf-8 -*-
"""usage: python tools/assemble_profiles.py <round tag>
Files the results of `tools/evidence.sh <round tag>` (gpurun_out/<round>/, the end-of-round state of the code) under
profiles/<round>_*: every bench line (the last JSON line of the file), the rocprofv3 kernel summaries, the scaling projection,
the parity report, the traces and timing tables; plus whatever a round's tools/evidence_extra.sh left there as extra_*."""
import os
import sys

RND = sys.argv[1] if len(sys.argv) > 1 else sys.exit(__doc__)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', RND) + '/'
P = os.path.join(ROOT, 'profiles') + '/'

NAMES = {'kernel_stats_c4.csv': 'bench_c4_kernel_stats.csv', 'kernel_stats_c3_zi.csv': 'zigap_c3_kernel_stats.csv',
         'kernel_stats_c5_sparse.csv': 'sparsegap_c5_kernel_stats.csv', 'zi_trace_nmf.txt': 'zigap_slow_path_trace_nmf.txt',
         'bench_c4_eighth_rccl1.json': 'bench_c4_eighth_rccl_one_rank.json'}
KEEP = ('bench_', 'kernel_stats_', 'scaling_projection', 'parity_errors', 'zi_trace_nmf', 'perf_', 'extra_', 'pytest_gpu')
for a in sorted(os.listdir(F)):
    if not a.startswith(KEEP) or a.endswith('.err') or os.path.isdir(F + a):
        continue
    text = open(F + a, 'rb').read()
    if a.endswith('.json') and a.startswith('bench_'):
        lines = [l for l in text.strip().splitlines() if l.startswith(b'{')]
        if not lines:
            continue
        text = lines[-1] + b'\n'
    if a.endswith('.txt'):
        text = b'\n'.join(l for l in text.splitlines() if b'amdgpu.ids' not in l) + b'\n'
    b = RND + '_' + NAMES.get(a, a[len('extra_'):] if a.startswith('extra_') else a)
    open(P + b, 'wb').write(text)
    print(b)

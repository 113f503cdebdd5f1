# -*- coding: utf-8 -*-
"""Copies the results of tools/evidence_r02.sh (gpurun_out/final/) into profiles/r02_* and rebuilds the two assembled
counter files (r02_sq_pass_c4.json, r02_pmc_hbm_c4.json) from that run's rocprofv3 --pmc passes."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'final') + '/'
P = os.path.join(ROOT, 'profiles') + '/'

old = json.load(open(P + 'r02_sq_pass_c4.json'))
raw = {}
for f in ('c4_sq1', 'c4_sq2', 'c4_sq3'):
    d = json.load(open(F + f + '.json'))['per_dispatch_mean']
    for k, dd in d.items():
        if 'fixup' not in k:
            raw.setdefault(k, {}).update({c: v for c, v in dd.items() if c != '_dispatches'})
new = dict(old)
new['round2_kernels'] = {'k_row_pass_k100': raw['k_row_pass_k100'], 'k_col_pass2': raw['k_col_pass2']}
der = dict(old['derived'])
for name in ('k_row_pass_k100', 'k_col_pass2'):
    n = raw[name]
    e = dict(der[name])
    # (slot counts of the workload: from the instruction counts per 16 slots of the first collection, same kernels)
    slots16 = old['round2_kernels'].get(name, raw[name])['SQ_INSTS_VALU'] / e['valu_instructions_per_16_slots']
    e['valu_instructions_per_16_slots'] = n['SQ_INSTS_VALU'] / slots16
    e['lds_instructions_per_16_slots'] = n['SQ_INSTS_LDS'] / slots16
    e['valu_busy_cycles_per_simd'] = n['SQ_ACTIVE_INST_VALU'] * 4 / 1024
    e['valu_busy_fraction'] = e['valu_busy_cycles_per_simd'] / (n['SQ_BUSY_CYCLES'] / 32)
    e['lds_bank_conflict_fraction'] = n['SQ_LDS_BANK_CONFLICT'] / n['SQ_LDS_IDX_ACTIVE']
    e['wave_cycles_waiting'] = n['SQ_WAIT_ANY'] / n['SQ_WAVE_CYCLES']
    e['wave_cycles_issuing'] = n['SQ_ACTIVE_INST_ANY'] / n['SQ_WAVE_CYCLES']
    e['wave_cycles_issue_stalled'] = 1.0 - e['wave_cycles_waiting'] - e['wave_cycles_issuing']
    der[name] = e
new['derived'] = der
json.dump(new, open(P + 'r02_sq_pass_c4.json', 'w'), indent=1)

h = json.load(open(P + 'r02_pmc_hbm_c4.json'))
fe = json.load(open(F + 'c4_fetch.json'))['per_dispatch_mean']
wr = json.load(open(F + 'c4_write.json'))['per_dispatch_mean']
h['counters'] = {'k_row_pass_k100': {'FETCH_SIZE': fe['k_row_pass_k100']['FETCH_SIZE'], 'WRITE_SIZE': wr['k_row_pass_k100']['WRITE_SIZE']},
                 'k_col_pass2': {'FETCH_SIZE': fe['k_col_pass2']['FETCH_SIZE'], 'WRITE_SIZE': wr['k_col_pass2']['WRITE_SIZE']},
                 'k_fixup': {'FETCH_SIZE': fe['oriana::k_fixup']['FETCH_SIZE'], 'WRITE_SIZE': wr['oriana::k_fixup']['WRITE_SIZE']}}
KiB = 1024.0
row = (2 * fe['k_row_pass_k100']['FETCH_SIZE'] + wr['k_row_pass_k100']['WRITE_SIZE']) * KiB
col = (fe['k_col_pass2']['FETCH_SIZE'] + wr['k_col_pass2']['WRITE_SIZE']) * KiB
fx = (fe['oriana::k_fixup']['FETCH_SIZE'] + wr['oriana::k_fixup']['WRITE_SIZE']) * KiB
h['traffic_bytes_per_pass'] = {'row_pass': row, 'col_pass': col, 'fixup': fx, 'total': row + col + fx}
json.dump(h, open(P + 'r02_pmc_hbm_c4.json', 'w'), indent=1)

cp = {'bench_c4.json': 'r02_bench_c4.json', 'bench_c2.json': 'r02_bench_c2.json', 'bench_c3_zi.json': 'r02_bench_c3_zi.json',
      'bench_c5_sparse.json': 'r02_bench_c5_sparse.json', 'bench_c5_sparse_split.json': 'r02_bench_c5_sparse_split.json',
      'bench_c4_eighth.json': 'r02_bench_c4_eighth.json', 'bench_c4_r1kernels.json': 'r02_bench_c4_r1kernels.json',
      'bench_c4_eighth_r1kernels.json': 'r02_bench_c4_eighth_r1kernels.json',
      'bench_c3_zi_f32mfma.json': 'r02_bench_c3_zi_f32mfma.json', 'bench_c3_zi_float64.json': 'r02_bench_c3_zi_float64.json',
      'kernel_stats.csv': 'r02_bench_c4_kernel_stats.csv', 'kernel_stats_c4_eighth.csv': 'r02_bench_c4_eighth_kernel_stats.csv',
      'kernel_stats_c3_zi.csv': 'r02_zigap_c3_kernel_stats.csv', 'kernel_stats_c5_sparse.csv': 'r02_sparsegap_c5_kernel_stats.csv',
      'parity_errors.json': 'r02_parity_errors.json', 'ubench_mfma_f32.txt': 'r02_ubench_mfma_f32.txt',
      'ubench_mfma_valu_overlap.txt': 'r02_ubench_mfma_valu_overlap.txt', 'ubench_mfma_bf16x3.txt': 'r02_ubench_mfma_bf16x3.txt',
      'ubench_lane_row.txt': 'r02_ubench_lane_row.txt', 'zi_a.json': 'r02_zigap_c3_sq_a.json', 'zi_b.json': 'r02_zigap_c3_sq_b.json'}
for a, b in cp.items():
    if os.path.exists(F + a):
        shutil.copy(F + a, P + b)
open(P + 'r02_zigap_c3_dense_kernels.txt', 'w').write(''.join(l for l in open(F + 'perf_zi_dense.txt') if 'amdgpu.ids' not in l))
print('traffic per pass: %.1f GB' % ((row + col + fx) / 1e9))

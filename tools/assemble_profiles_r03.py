# -*- coding: utf-8 -*-
"""Files the results of tools/evidence_r03.sh / evidence_r03_pmc.sh (gpurun_out/r03/) under profiles/r03_* and assembles
the counter files (r03_sq_pass_c4_hybrid.json, r03_pmc_hbm_c4_hybrid.json) from that run's rocprofv3 --pmc passes."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'r03') + '/'
P = os.path.join(ROOT, 'profiles') + '/'

cp = {'bench_c4.json': 'r03_bench_c4.json', 'bench_c4_sliced.json': 'r03_bench_c4_sliced.json', 'bench_c2.json': 'r03_bench_c2.json',
      'bench_c3_zi.json': 'r03_bench_c3_zi.json', 'bench_c5_sparse.json': 'r03_bench_c5_sparse.json',
      'bench_c4_eighth.json': 'r03_bench_c4_eighth.json', 'bench_c4_eighth_sliced.json': 'r03_bench_c4_eighth_sliced.json',
      'bench_c4_eighth_z05.json': 'r03_bench_c4_eighth_z05.json', 'bench_c4_eighth_z05_sliced.json': 'r03_bench_c4_eighth_z05_sliced.json',
      'kernel_stats_c4.csv': 'r03_bench_c4_kernel_stats.csv', 'kernel_stats_c4_eighth.csv': 'r03_bench_c4_eighth_kernel_stats.csv',
      'kernel_stats_c3_zi.csv': 'r03_zigap_c3_kernel_stats.csv', 'kernel_stats_c5_sparse.csv': 'r03_sparsegap_c5_kernel_stats.csv',
      'parity_errors.json': 'r03_parity_errors.json', 'ubench_mfma_mix.txt': 'r03_ubench_mfma_mix.txt',
      'dense_row_stamps.txt': 'r03_dense_row_stamps.txt', 'dense_row_ablations.txt': 'r03_dense_row_ablations.txt'}
for a, b in cp.items():
    if os.path.exists(F + a):
        shutil.copy(F + a, P + b)

# threshold sweep of the hybrid layout at full C4
rows = [json.loads(l) for l in open(F + 'dense_threshold_c4.txt') if l.startswith('{')]
json.dump({'command': 'python3 tools/perf_dense_pass.py 1000000 30000 100 0.1 0 0.3 0.25 0.2 0.15 0.1',
           'what': 'the responsibility pass (HIP events per kernel, mean of 5 after 2 warm-ups) on the c4-like synthetic matrix for several '
                   'density thresholds of the hybrid layout; threshold 0 = sliced layout only; vs_first = max column-relative difference '
                   'of Z_i, Z_j from the sliced layout\'s', 'runs': rows}, open(P + 'r03_dense_threshold_c4.json', 'w'), indent=1)

sq = {}
for f in ('r03_c4_sq1', 'r03_c4_sq2'):
    d = json.load(open(F + f + '.json'))['per_dispatch_mean']
    for k, dd in d.items():
        if any(t in k for t in ('k_row_pass', 'k_col_pass', 'k_dn_row', 'k_dn_col')):
            sq.setdefault(k, {}).update({c: v for c, v in dd.items() if c != '_dispatches'})
der = {}
for k, n in sq.items():
    clk = n['GRBM_GUI_ACTIVE'] / 8.0                       # cycles of the kernel (the counter sums over the 8 XCDs)
    e = {'kernel_cycles': clk, 'valu_busy_fraction': n['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / clk,
         'wave_cycles_waiting': n['SQ_WAIT_ANY'] / n['SQ_WAVE_CYCLES'], 'wave_cycles_issue_stalled': n['SQ_WAIT_INST_ANY'] / n['SQ_WAVE_CYCLES'],
         'wave_cycles_issuing': n['SQ_ACTIVE_INST_ANY'] / n['SQ_WAVE_CYCLES'],
         'lds_bank_conflict_fraction': n['SQ_LDS_BANK_CONFLICT'] / max(n['SQ_LDS_IDX_ACTIVE'], 1.0)}
    if n.get('SQ_INSTS_MFMA'):
        e['matrix_pipe_busy_fraction'] = n['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / clk
        e['valu_instructions_per_matrix_instruction'] = n['SQ_INSTS_VALU'] / n['SQ_INSTS_MFMA']
        e['lds_instructions_per_matrix_instruction'] = n['SQ_INSTS_LDS'] / n['SQ_INSTS_MFMA']
    der[k] = e
json.dump({'command': 'tools/evidence_r03_pmc.sh: two rocprofv3 --pmc passes (8 SQ counters + GRBM_GUI_ACTIVE, counters only with --kernel-trace) over '
                      'python3 bench.py --steps 2 --warmup 1 --no-cpu (c4, hybrid layout, threshold 0.2)',
           'unit': 'per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles',
           'counters': sq, 'derived': der}, open(P + 'r03_sq_pass_c4_hybrid.json', 'w'), indent=1)

fe = json.load(open(F + 'r03_c4_fetch.json'))['per_dispatch_mean']
wr = json.load(open(F + 'r03_c4_write.json'))['per_dispatch_mean']
KiB = 1024.0


def g(d, key):
    for k, v in d.items():
        if key in k:
            return v[[c for c in v if c != '_dispatches'][0]]
    return 0.0


tr = {'row_pass': (2 * g(fe, 'k_row_pass') + g(wr, 'k_row_pass')) * KiB, 'col_pass': (g(fe, 'k_col_pass') + g(wr, 'k_col_pass')) * KiB,
      'dense_row': (2 * g(fe, 'k_dn_row') + g(wr, 'k_dn_row')) * KiB, 'dense_col': (2 * g(fe, 'k_dn_col') + g(wr, 'k_dn_col')) * KiB,
      'dense_images': (g(fe, 'k_dn_images<6, 1, true>') + g(fe, 'k_dn_images<6, 1, false>') + g(wr, 'k_dn_images<6, 1, true>') + g(wr, 'k_dn_images<6, 1, false>')) * KiB,
      'fixup': (g(fe, 'k_fixup') + g(fe, 'k_dn_fixup') + g(wr, 'k_fixup')) * KiB}
tr['total'] = sum(tr.values())
names = ('k_row_pass', 'k_col_pass', 'k_dn_row', 'k_dn_col', 'k_dn_images<6, 1, true>', 'k_dn_images<6, 1, false>', 'k_fixup', 'k_dn_fixup')
json.dump({'command': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu   (and a second, '
                      'separate pass with --pmc WRITE_SIZE); tools/evidence_r03_pmc.sh', 'workload': 'c4',
           'layout': 'hybrid, threshold 0.2 (4064 dense genes)', 'n_gpus': 1,
           'unit': 'KB per launch, averaged over the launches of the run, as rocprofv3 reports them',
           'counters': {k: {'FETCH_SIZE': g(fe, k), 'WRITE_SIZE': g(wr, k)} for k in names},
           'corrections': 'MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read.  Doubled: the sliced row pass '
                          '(8-byte records as 16-byte per-lane loads), the dense row kernel (counts as 16-byte per-lane loads, operand images by 16-byte LDS-DMA) and the '
                          'dense gene-side kernel (s and the operand images by 16-byte LDS-DMA; check: 4.06e9 entries x 4 B = 16.3 GB of s against 9.67 GB counted).  '
                          'The sliced column pass reads 4-byte and 1-byte per-lane streams plus factor tiles served by L2 / Infinity Cache: taken as counted.  '
                          'WRITE_SIZE as counted; KB taken as KiB.',
           'traffic_bytes_per_pass': tr, 'algorithmic_bytes': 120824000000.0,
           'note': 'sliced layout alone (profiles/r02_pmc_hbm_c4.json): 72.2 GB; the hybrid layout adds the round trip of s for the dense entries (4 B written, 4 B read) and '
                   'their uint16 counts'}, open(P + 'r03_pmc_hbm_c4_hybrid.json', 'w'), indent=1)
print('traffic per pass: %.1f GB' % (tr['total'] / 1e9), {k: round(v / 1e9, 1) for k, v in tr.items()})
for k, e in der.items():
    print(k[:34], {a: round(b, 3) for a, b in e.items()})

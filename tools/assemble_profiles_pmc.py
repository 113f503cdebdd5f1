# -*- coding: utf-8 -*-
"""usage: python tools/assemble_profiles_pmc.py <round tag>
Files the counter passes of `tools/evidence_pmc.sh <round tag>` (gpurun_out/<round>/<round>_<workload>_{sq1,sq2,fetch,write}.json)
under profiles/: <round>_sq_secondary.json (what binds the kernels of configs[2] / configs[4]), <round>_pmc_hbm_<workload>.json (HBM
traffic per sweep, with the guide's gfx950 correction spelled out per kernel -- the `roofline.traffic` of those workloads),
<round>_pmc_hbm_c4_hybrid.json and <round>_sq_pass_c4_hybrid.json for the headline.  Every file records the command that ran."""
import json
import os
import sys

RND = sys.argv[1] if len(sys.argv) > 1 else 'r04'            # the round's tag: gpurun_out/<RND>/<RND>_*.json -> profiles/<RND>_*.json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', RND) + '/'
P = os.path.join(ROOT, 'profiles') + '/'
KiB = 1024.0
ALG = {'c3_zi': 16.0 * 100000 * 20000 + 4.0 * 50 * (2 * 100000 + 2 * 20000) + 4.0 * 50 * 20000,
       'c5_sparse': 4.0 * 500000 * 25000 + 4.0 * 64 * (2 * 500000 + 2 * 25000) + 12.0 * 64 * 25000}
# kernels whose reads are wide coalesced streams (16 B per lane, global_load or LDS-DMA): FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)
WIDE = ('k_row_pass', 'k_dt_times_factor', 'k_zi_row', 'k_zi_col', 'k_dropout_sweep')
SKIP = ('k_dropout_fused', 'k_dropout_fix_nz')         # the float64 A/B leg of bench.py and the one-time initialisation: not part of a sweep

sq_all = {}
for w in ('c3_zi', 'c5_sparse'):
    try:
        s1 = json.load(open(F + '%s_%s_sq1.json' % (RND, w)))['per_dispatch_mean']
        s2 = json.load(open(F + '%s_%s_sq2.json' % (RND, w)))['per_dispatch_mean']
        fe = json.load(open(F + '%s_%s_fetch.json' % (RND, w)))['per_dispatch_mean']
        wr = json.load(open(F + '%s_%s_write.json' % (RND, w)))['per_dispatch_mean']
    except FileNotFoundError:
        continue
    der = {}
    for k in s1:
        if any(x in k for x in SKIP) or k not in s2:
            continue
        n = dict(s1[k]); n.update(s2[k])
        clk = n['GRBM_GUI_ACTIVE'] / 8.0
        e = {'kernel_cycles': clk, 'valu_busy_fraction': n['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / clk,
             'wave_cycles_waiting': n['SQ_WAIT_ANY'] / n['SQ_WAVE_CYCLES'],
             'wave_cycles_issue_stalled': n['SQ_WAIT_INST_ANY'] / n['SQ_WAVE_CYCLES'],
             'wave_cycles_issuing': n['SQ_ACTIVE_INST_ANY'] / n['SQ_WAVE_CYCLES'],
             'lds_bank_conflict_fraction': n['SQ_LDS_BANK_CONFLICT'] / max(n['SQ_LDS_IDX_ACTIVE'], 1.0),
             'valu_instructions': n['SQ_INSTS_VALU'], 'lds_instructions': n['SQ_INSTS_LDS']}
        if n.get('SQ_INSTS_MFMA'):
            e['matrix_pipe_busy_fraction'] = n['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / clk
            e['valu_instructions_per_matrix_instruction'] = n['SQ_INSTS_VALU'] / n['SQ_INSTS_MFMA']
        der[k] = e
    sq_all[w] = der
    tr, raw = {}, {}
    for k in fe:
        if any(x in k for x in SKIP):
            continue
        f = fe[k].get('FETCH_SIZE', 0.0)
        wv = wr.get(k, {}).get('WRITE_SIZE', 0.0)
        raw[k] = {'FETCH_SIZE': f, 'WRITE_SIZE': wv}
        tr[k] = ((2.0 if any(x in k for x in WIDE) else 1.0) * f + wv) * KiB
    tr['total'] = sum(tr.values())
    json.dump({'command': ('rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --workload %s --steps 3 --warmup 1 --no-cpu '
                           '(and a second, separate pass with --pmc WRITE_SIZE); tools/evidence_pmc.sh %s') % (w, RND),
               'workload': w, 'n_gpus': 1, 'unit': 'KB per launch (one launch of each kernel per sweep), as rocprofv3 reports them',
               'counters': raw,
               'corrections': 'MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read.  Doubled: '
                              + ', '.join(WIDE) + ' (16-byte per-lane record / D_hat streams, LDS-DMA).  The column passes read 8-byte + 2-byte per-lane '
                              'streams plus factor tiles served by L2 / Infinity Cache: taken as counted.  WRITE_SIZE as counted; KB taken as KiB.  Left out: '
                              + ', '.join(SKIP) + ' (the float64 A/B leg of the bench line and the one-time initialisation).',
               'traffic_bytes_per_pass': tr, 'algorithmic_bytes': ALG[w]}, open(P + '%s_pmc_hbm_%s.json' % (RND, w), 'w'), indent=1)
    print(w, 'traffic per sweep %.1f GB against %.1f GB algorithmic' % (tr['total'] / 1e9, ALG[w] / 1e9), {k[:28]: round(v / 1e9, 2) for k, v in tr.items()})
if sq_all:
  json.dump({'command': 'tools/evidence_pmc.sh ' + RND + ': two rocprofv3 --pmc passes per workload (8 SQ counters + GRBM_GUI_ACTIVE; counters only with --kernel-trace) over '
                      'python3 bench.py --workload {c3_zi, c5_sparse} --steps 3 --warmup 1 --no-cpu',
           'unit': 'per launch; wave-cycle shares of SQ_WAVE_CYCLES; busy fractions of the kernel\'s cycles x 1024 SIMDs',
           'derived': sq_all}, open(P + RND + '_sq_secondary.json', 'w'), indent=1)
for w, der in sq_all.items():
    for k, e in der.items():
        print(w, k[:40], {a: round(b, 3) for a, b in e.items() if 'fraction' in a or 'wave_cycles' in a})

# ---- C4, hybrid layout, final build
try:
    fe = json.load(open(F + RND + '_c4_fetch.json'))['per_dispatch_mean']
    wr = json.load(open(F + RND + '_c4_write.json'))['per_dispatch_mean']
    s1 = json.load(open(F + RND + '_c4_sq1.json'))['per_dispatch_mean']
    s2 = json.load(open(F + RND + '_c4_sq2.json'))['per_dispatch_mean']
except FileNotFoundError:
    fe = None
if fe is not None:
    def g(d, key):
        for k, v in d.items():
            if key in k:
                return v[[c for c in v if c != '_dispatches'][0]]
        return 0.0
    tr = {'row_pass': (2 * g(fe, 'k_row_pass') + g(wr, 'k_row_pass')) * KiB, 'col_pass': (g(fe, 'k_col_pass') + g(wr, 'k_col_pass')) * KiB,
          'dense_row': (2 * g(fe, 'k_dn_row') + g(wr, 'k_dn_row')) * KiB, 'dense_col': (2 * g(fe, 'k_dn_col') + g(wr, 'k_dn_col')) * KiB,
          'dense_images': (g(fe, 'k_dn_images<6, 1, true>') + g(fe, 'k_dn_images<6, 1, false>') + g(wr, 'k_dn_images<6, 1, true>')
                           + g(wr, 'k_dn_images<6, 1, false>')) * KiB,
          'fixup': (g(fe, 'k_fixup') + g(fe, 'k_dn_fixup') + g(wr, 'k_fixup')) * KiB}
    tr['total'] = sum(tr.values())
    names = ('k_row_pass', 'k_col_pass', 'k_dn_row', 'k_dn_col', 'k_dn_images<6, 1, true>', 'k_dn_images<6, 1, false>', 'k_fixup', 'k_dn_fixup')
    json.dump({'command': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu   (and a second, '
                          'separate pass with --pmc WRITE_SIZE); tools/evidence_pmc.sh ' + RND, 'workload': 'c4',
               'layout': 'hybrid, threshold 0.2 (4064 dense genes); the row blocks of the last round of both row kernels split into three gene ranges',
               'n_gpus': 1, 'unit': 'KB per launch, averaged over the launches of the run, as rocprofv3 reports them',
               'counters': {k: {'FETCH_SIZE': g(fe, k), 'WRITE_SIZE': g(wr, k)} for k in names},
               'corrections': 'MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read.  Doubled: the sliced '
                              'row pass (8-byte records as 16-byte per-lane loads), the dense row kernel (counts as 16-byte per-lane loads, operand images by '
                              '16-byte LDS-DMA) and the dense gene-side kernel (s and the operand images by 16-byte LDS-DMA).  The sliced column pass reads 4-byte '
                              'and 1-byte per-lane streams plus factor tiles served by L2 / Infinity Cache: taken as counted.  WRITE_SIZE as counted; KB taken as KiB.',
               'traffic_bytes_per_pass': tr, 'algorithmic_bytes': 120824000000.0}, open(P + RND + '_pmc_hbm_c4_hybrid.json', 'w'), indent=1)
    sq, der = {}, {}
    for d in (s1, s2):
        for k, dd in d.items():
            if any(t in k for t in ('k_row_pass', 'k_col_pass', 'k_dn_row', 'k_dn_col')):
                sq.setdefault(k, {}).update({c: v for c, v in dd.items() if c != '_dispatches'})
    for k, n in sq.items():
        clk = n['GRBM_GUI_ACTIVE'] / 8.0
        e = {'kernel_cycles': clk, 'valu_busy_fraction': n['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / clk,
             'wave_cycles_waiting': n['SQ_WAIT_ANY'] / n['SQ_WAVE_CYCLES'], 'wave_cycles_issue_stalled': n['SQ_WAIT_INST_ANY'] / n['SQ_WAVE_CYCLES'],
             'wave_cycles_issuing': n['SQ_ACTIVE_INST_ANY'] / n['SQ_WAVE_CYCLES'],
             'lds_bank_conflict_fraction': n['SQ_LDS_BANK_CONFLICT'] / max(n['SQ_LDS_IDX_ACTIVE'], 1.0)}
        if n.get('SQ_INSTS_MFMA'):
            e['matrix_pipe_busy_fraction'] = n['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / clk
            e['valu_instructions_per_matrix_instruction'] = n['SQ_INSTS_VALU'] / n['SQ_INSTS_MFMA']
        der[k] = e
    json.dump({'command': 'tools/evidence_pmc.sh ' + RND + ': two rocprofv3 --pmc passes (8 SQ counters + GRBM_GUI_ACTIVE, counters only with --kernel-trace) over '
                          'python3 bench.py --steps 2 --warmup 1 --no-cpu (c4, hybrid layout, final build)',
               'unit': 'per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles',
               'counters': sq, 'derived': der}, open(P + RND + '_sq_pass_c4_hybrid.json', 'w'), indent=1)
    print('c4 hybrid traffic per pass %.1f GB' % (tr['total'] / 1e9), {k: round(v / 1e9, 2) for k, v in tr.items()})
    for k, e in der.items():
        print(k[:40], {a: round(b, 3) for a, b in e.items() if b is not None and ('fraction' in a or 'wave_cycles' in a)})

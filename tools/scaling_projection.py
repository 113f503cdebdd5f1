# -*- coding: utf-8 -*-
"""A PROJECTED 1 / 2 / 4 / 8-GPU table for the headline configuration (1M x 30k, K = 100) from ONE-GPU measurements of the
shares a rank would own (bench.py --workload c4 | c4_half | c4_quarter | c4_eighth).  No multi-GPU node has been available in
any round: every figure below the one-GPU rows is a projection, labelled as such, never a measurement.

    python tools/scaling_projection.py c4.json c4_half.json c4_quarter.json c4_eighth.json [out.json]

The shares should be measured with ORIANA_BENCH_FORCE_PG=1 (tools/evidence.sh does): the sweep then runs the sharded code path
over a one-rank nccl group -- packed Z_j order, the extra gene-side finalize, the exchange's synchronisation points -- so that only
the wire time of the collectives is modelled.  Per share: the step time, the kernels of the pass, the fixed part (cell- and gene-side updates, M-step, preparation, launch
gaps = step - pass) -- all measured.  The exchange of a sharded sweep: one float64 all-reduce of the
rate partials (2K doubles, started before the column pass), the float32 all-reduce of the per-gene sums Z_j in two
segments -- the sliced genes' segment (m - gd rows) travels under the dense gene-side kernel, the dense genes' gd rows
follow it.  Its time is MODELLED: ring all-reduce, 2 (N - 1) / N x bytes over ONE xGMI link (153 GB/s peak per link and
direction, MI355X_MICROARCH.md; 70 % of it assumed) + 25 us per collective -- RCCL would use several of the 7 links, so
this is the pessimistic end.  'exposed': nothing hidden (all three collectives on the critical path); 'hidden': only the
dense genes' segment and one latency exposed (what engine.zq_gap(on_segment=...) arranges)."""
import json
import sys

LINK_GBS = 153.0
LINK_EFF = 0.7
LAT_US = 25.0


def load(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith('{')]
    return json.loads(lines[-1])


def ring_ms(nbytes, N):
    if N <= 1:
        return 0.0
    return 2.0 * (N - 1) / N * nbytes / (LINK_GBS * LINK_EFF * 1e9) * 1e3 + LAT_US * 1e-3


def main():
    paths = sys.argv[1:5]
    out_path = sys.argv[5] if len(sys.argv) > 5 else None
    rows = []
    base = None
    for N, p in zip((1, 2, 4, 8), paths):
        d = load(p)
        cfg = d['config']
        m, K = cfg['n_genes'], cfg['K']
        ks = d['roofline']['kernel_ms']
        pass_ms = sum(ks.values())
        step = d['ms_per_step']
        layout = cfg.get('layout', '')
        gd = int(layout.split('hybrid: ')[1].split(' genes')[0]) if layout.startswith('hybrid') else 0
        f32_bytes = 4.0 * m * K
        dense_bytes = 4.0 * gd * K
        f64_bytes = 8.0 * 2 * K
        exposed = ring_ms(f32_bytes - dense_bytes, N) + ring_ms(dense_bytes, N) + ring_ms(f64_bytes, N) if N > 1 else 0.0
        hidden = ring_ms(dense_bytes, N) if N > 1 else 0.0
        sharded = 'exchange_rehearsal' in d        # ORIANA_BENCH_FORCE_PG=1: the share ran the SHARDED sweep over a one-rank nccl group
        row = {'gpus': N, 'rows_per_rank': cfg['rows_per_rank'],
               'measured_on': ('1 GPU, this share alone, through the sharded code path (one-rank nccl group: packed Z_j order, gene-side '
                               'finalize, every collective issued as a self all-reduce)' if sharded else
                               '1 GPU (this share alone, unsharded code path: the packed exchange order and its extra gene-side launch '
                               'are NOT in the step time)'),
               'step_ms': step, 'pass_ms': pass_ms, 'fixed_ms': step - pass_ms, 'kernel_ms': ks,
               'exchange_bytes_f32': f32_bytes, 'exchange_bytes_f32_dense_segment': dense_bytes, 'exchange_bytes_f64': f64_bytes,
               'exchange_ms_exposed_model': exposed, 'exchange_ms_hidden_model': hidden,
               'sweeps_per_s_exposed': 1e3 / (step + exposed), 'sweeps_per_s_hidden': 1e3 / (step + hidden),
               'roofline_frac_of_share': d['roofline']['frac']}
        if N == 1:
            base = row['sweeps_per_s_hidden']
            row['note'] = 'MEASURED: the headline run itself'
        else:
            row['note'] = ('PROJECTION: one rank\'s share measured on one GPU + the modelled exchange; the ranks are assumed to take '
                           'the same time (the generator\'s rows are exchangeable)')
        row['speedup_hidden'] = row['sweeps_per_s_hidden'] / base
        row['efficiency_hidden'] = row['speedup_hidden'] / N
        row['speedup_exposed'] = row['sweeps_per_s_exposed'] / base
        rows.append(row)
    out = {'what': 'PROJECTED scaling of configs[3] (1M x 30k, K = 100) at 1 / 2 / 4 / 8 MI355X from one-GPU measurements of each '
                   'rank\'s share; NOT a measurement of a multi-GPU run (no 8-GPU node has been available in any round)',
           'exchange_model': {'link_GB_per_s': LINK_GBS, 'assumed_efficiency': LINK_EFF, 'latency_us_per_collective': LAT_US,
                              'algorithm': 'ring all-reduce over one xGMI link: 2 (N - 1) / N x bytes / bandwidth + latency'},
           'rows': rows}
    txt = json.dumps(out, indent=1)
    if out_path:
        with open(out_path, 'w') as f:
            f.write(txt + '\n')
    print('| GPUs | rows per rank | step (ms) | pass (ms) | fixed (ms) | exchange exposed / hidden (ms, model) | sweeps/s exposed / hidden | speed-up (hidden) | efficiency |')
    print('|---|---|---|---|---|---|---|---|---|')
    for r in rows:
        print('| %d | %d | %.2f | %.2f | %.2f | %.3f / %.3f | %.1f / %.1f | %.2f | %.2f |' % (
            r['gpus'], r['rows_per_rank'], r['step_ms'], r['pass_ms'], r['fixed_ms'], r['exchange_ms_exposed_model'],
            r['exchange_ms_hidden_model'], r['sweeps_per_s_exposed'], r['sweeps_per_s_hidden'], r['speedup_hidden'], r['efficiency_hidden']))


if __name__ == '__main__':
    main()

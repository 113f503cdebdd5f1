"""Achieved parity of the HIP path against the reference's golden states (tests/golden/*.npz), per key:
  colrel = max |got - ref| / (|ref| + colmax|ref|)            (the metric of tests/helpers.py)
  strict = max |got - ref| / |ref| over the entries with |ref| > 1e-12 * colmax|ref| (element-wise relative error)
  abs    = max |got - ref|
for every single sweep s_a -> s_b started from the reference's own state, on every model / shape / start.
And the same sweeps against the EXACT yardstick (oracle/cavi_oracle.py with exact = True: the loop nest in float64 from
the same float32 inputs, everything else as the reference): `vs_exact` holds, per key, max |reference - exact| and
max |HIP - exact| in the key's metric (absolute for the Bernoulli posteriors, colrel otherwise) -- the reference's
float32 loop nest is itself 1e-7 .. 2e-6 away from exact, so "no further from exact than the reference is" is the
meaningful bound for the quantities whose conditioning amplifies rounding (p_s, S_hat, pi_s).
Writes profiles-style JSON to the path given (default gpurun_out/parity_errors.json).  Needs the GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import golden_files, load_golden, state_of, PARAM_KEYS, EXPECT_KEYS   # noqa: E402
import oriana_amd.models as M   # noqa: E402
from oracle import cavi_oracle as co   # noqa: E402

ABS_KEYS = ('p_d', 'pi_d', 'p_s', 'pi_s', 'S_hat')
ORACLE = {'GaP': co.OracleGaP, 'ZIGaP': co.OracleZIGaP, 'SparseGaP': co.OracleSparseGaP, 'SparseZIGaP': co.OracleSparseZIGaP}


def errs(got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    cm = np.abs(ref).max(axis=0, keepdims=True) if ref.ndim == 2 else np.abs(ref).max()
    d = np.abs(got - ref)
    fin = np.isfinite(d)
    colrel = float((d / (np.abs(ref) + cm + 1e-300))[fin].max()) if fin.any() else 0.0
    big = fin & (np.abs(ref) > 1e-12 * (cm + 1e-300))
    strict = float((d[big] / np.abs(ref)[big]).max()) if big.any() else 0.0
    return colrel, strict, float(d[fin].max()) if fin.any() else 0.0


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'parity_errors.json')
    per_file = {}
    worst = {}
    vs_exact = {}
    for path in golden_files():
        g = load_golden(path)
        name = str(g['meta/name'])
        cls = getattr(M, name)
        model = cls(g['X'], k=int(g['meta/k']), use_factors=bool(g['meta/use_factors']), tau=float(g['meta/tau']),
                    init=(g['s0/a1'], g['s0/b1']))
        rec = {}
        ex = ORACLE[name](g['X'], int(g['meta/k']), g['s0/a1'], g['s0/b1'], tau=float(g['meta/tau']))
        ex.exact = True
        for a, b in (('s0', 's1'), ('s1', 's2'), ('s2', 's3')):
            model.load_state(state_of(g, a))
            model.step()
            got, ref = model.state(), state_of(g, b)
            ex.load_state(state_of(g, a))
            if ex.zi:
                ex.D_hat = co.bernoulli_mean(ex.p_d)      # (the golden states hold p_d; its float32 cast is an expectation)
            ex.step()
            exact = ex.state()
            # (the yardstick is float64: it does not underflow where float32 does, so only the well-conditioned starts --
            #  use_factors=False, the `rand` goldens -- are compared against it)
            for k in (PARAM_KEYS + EXPECT_KEYS if not bool(g['meta/use_factors']) else []):
                if k in got and k in ref and k in exact:
                    i = 2 if k in ABS_KEYS else 0
                    e_ref, e_hip = errs(ref[k], exact[k])[i], errs(got[k], exact[k])[i]
                    v = vs_exact.setdefault(k, {'metric': 'abs' if k in ABS_KEYS else 'colrel', 'reference_minus_exact': 0.0,
                                                'hip_minus_exact': 0.0, 'worst_ratio_hip_over_reference': 0.0})
                    v['reference_minus_exact'] = max(v['reference_minus_exact'], e_ref)
                    v['hip_minus_exact'] = max(v['hip_minus_exact'], e_hip)
                    if e_hip > 1e-9 and e_ref > 0:
                        v['worst_ratio_hip_over_reference'] = max(v['worst_ratio_hip_over_reference'], e_hip / e_ref)
            for k in PARAM_KEYS + EXPECT_KEYS:
                if k in got and k in ref:
                    c, s, d = errs(got[k], ref[k])
                    r = rec.setdefault(k, {'colrel': 0.0, 'strict': 0.0, 'abs': 0.0})
                    r['colrel'] = max(r['colrel'], c); r['strict'] = max(r['strict'], s); r['abs'] = max(r['abs'], d)
        per_file[os.path.basename(path)] = rec
        for k, r in rec.items():
            w = worst.setdefault(k, {'colrel': 0.0, 'strict': 0.0, 'abs': 0.0})
            for f in r:
                w[f] = max(w[f], r[f])
    res = {'what': __doc__.strip().split('\n')[0], 'sweeps': 's0->s1, s1->s2, s2->s3 from the reference\'s own states',
           'worst_over_all_goldens': worst, 'vs_exact': vs_exact, 'per_golden': per_file}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, 'w'), indent=1, sort_keys=True)
    for k, w in sorted(worst.items()):
        print('%-10s colrel %.2e  strict %.2e  abs %.2e' % (k, w['colrel'], w['strict'], w['abs']))
    for k, v in sorted(vs_exact.items()):
        print('vs exact %-10s (%s)  reference %.2e  HIP %.2e  worst ratio %.2f' % (k, v['metric'], v['reference_minus_exact'], v['hip_minus_exact'], v['worst_ratio_hip_over_reference']))


if __name__ == '__main__':
    main()

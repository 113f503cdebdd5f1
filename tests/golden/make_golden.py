# -*- coding: utf-8 -*-
"""Generate the golden vectors under tests/golden/ from the reference implementation.

Runs ONLY in the build container (it imports /root/reference; the GPU box has no copy).
The outputs are data: inputs and the reference's outputs.  No reference source is stored.

The reference is imported as-is with two in-process shims for ordinary Python errors on
this image:
  * ``np.float`` / ``np.int`` aliases (removed from NumPy >= 1.24; parameters.py:11 etc.)
  * a stub ``numba`` module whose ``jit`` returns the function unchanged (numba is not
    installed), so the four loop nests run as plain NumPy-scalar Python.
``SparseGaP.step()`` raises NameError in the reference (sparse_gap.py:127 reads a bare
``S_hat``).  To capture its evident intent without restating any reference code, the name
``S_hat`` is injected into that module's globals (bound to ``model.S_hat``) before each
step; the fixture records ``sparsegap_patch='module-global S_hat = model.S_hat'``.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('ORIANA_REFERENCE', '/root/reference')


def import_reference():
    np.float = float      # shim 1
    np.int = int
    nb = types.ModuleType('numba')  # shim 2

    def jit(*a, **k):
        def deco(f):
            return f
        return deco
    nb.jit = jit
    sys.modules['numba'] = nb
    sys.path.insert(0, REF)
    import oriana  # noqa: F401
    import oriana.models
    import oriana.models.sparse_gap
    import oriana.singlecell
    import oriana.utils
    return oriana


PARAM_KEYS = ['alpha1', 'alpha2', 'beta1', 'beta2', 'a1', 'a2', 'b1', 'b2', 'pi_d', 'p_d', 'pi_s', 'p_s']
ARRAY_KEYS = {'U_hat': 'U_hat', 'V_hat': 'V_hat', 'Vprime_hat': 'V_hat', 'log_U_hat': 'log_U_hat',
              'log_V_hat': 'log_V_hat', 'log_Vprime_hat': 'log_V_hat', 'S_hat': 'S_hat'}


def snapshot(model, tag, out):
    for key in PARAM_KEYS:
        if hasattr(model, key):
            out['%s/%s' % (tag, key)] = np.array(getattr(model, key)[:])
    for key, name in ARRAY_KEYS.items():
        if hasattr(model, key):
            out['%s/%s' % (tag, name)] = np.array(getattr(model, key))


def kernel_io(oriana, name, model, out):
    """One raw call of the model's compute_Z_q_expectations on the post-init state."""
    n, m, K = model.n, model.m, model.k
    X = model.X[:].astype(np.float32)
    Zi = np.empty((n, K), dtype=np.float32)
    Zj = np.empty((m, K), dtype=np.float32)
    Zl = np.empty((m, K), dtype=np.float32)
    M = oriana.models
    if name == 'GaP':
        M.GaP.compute_Z_q_expectations(Zi, Zj, model.log_U_hat, model.log_V_hat, X)
        Zl[:] = 0
    elif name == 'ZIGaP':
        M.ZIGaP.compute_Z_q_expectations(Zi, Zj, Zl, model.log_U_hat, model.log_V_hat, model.D_hat, X)
    elif name == 'SparseGaP':
        St = (model.p_s[:] > model.tau).astype(np.float32)
        M.SparseGaP.compute_Z_q_expectations(Zi, Zj, Zl, model.log_U_hat, model.log_Vprime_hat,
                                             St, model.S_hat, X)
    else:
        St = (model.p_s[:] > model.tau).astype(np.float32)
        M.SparseZIGaP.compute_Z_q_expectations(Zi, Zj, Zl, model.log_U_hat, model.log_Vprime_hat,
                                               St, model.S_hat, model.D_hat, X)
    out['kernel/Zi'] = Zi
    out['kernel/Zj'] = Zj
    out['kernel/Zlog'] = Zl


def run_case(oriana, name, n, m, k, use_factors, seed, sweeps=(1, 2, 3, 10)):
    from oriana.singlecell import CountMatrix, generate_factor_matrices
    np.random.seed(seed)
    X, _, _, _ = generate_factor_matrices(n, m, k)
    np.random.seed(seed + 1)
    cls = getattr(oriana.models, name)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = cls(CountMatrix(np.asarray(X)), k=k, use_factors=use_factors)
    out = {'X': np.asarray(X, dtype=np.int64),
           'meta/name': np.array(name), 'meta/k': np.array(k), 'meta/use_factors': np.array(use_factors),
           'meta/seed': np.array(seed), 'meta/tau': np.array(getattr(model, 'tau', 0.5)),
           'meta/sparsegap_patch': np.array('module-global S_hat = model.S_hat' if name == 'SparseGaP' else ''),
           # NMF warm start as left in the U / V node buffers by base.py:38-40 (an INPUT fixture:
           # scikit-learn is unpinned, its output is not something to reproduce)
           'nmf/U': np.array(model.U[:]), 'nmf/V': np.array(model.V[:])}
    snapshot(model, 's0', out)
    kernel_io(oriana, name, model, out)
    done = 0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for target in sweeps:
            while done < target:
                if name == 'SparseGaP':
                    oriana.models.sparse_gap.S_hat = model.S_hat
                model.step()
                done += 1
            snapshot(model, 's%d' % target, out)
    return out


def metrics_case(oriana, n, m, k, seed, sweeps=3):
    """reconstruction_deviance / explained_deviance / frobenius_norm of the reference (base.py:58-87;
    only SparseZIGaP defines loglikelihood_X, sparse_zigap.py:44-51), called in the order of
    experiments/clustering.py:26-27, once with the generator's integer X (the per-entry terms are then
    truncated by `np.empty_like(X)`, sparse_zigap.py:45) and once with the same X as float64."""
    from oriana.singlecell import CountMatrix, generate_factor_matrices
    out = {}
    for tag, dtype in (('int', np.int64), ('float', np.float64)):
        np.random.seed(seed)
        X, _, _, _ = generate_factor_matrices(n, m, k)
        np.random.seed(seed + 1)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = oriana.models.SparseZIGaP(CountMatrix(np.asarray(X).astype(dtype)), k=k, use_factors=False)
            out['s0/a1'] = np.array(model.a1[:]); out['s0/b1'] = np.array(model.b1[:])
            for _ in range(sweeps):
                model.step()
            snapshot(model, 's%d' % sweeps, out)
            out['X'] = np.asarray(X, dtype=np.int64)
            out['metrics_%s/reconstruction_deviance' % tag] = np.array(model.reconstruction_deviance())
            out['metrics_%s/explained_deviance' % tag] = np.array(model.explained_deviance())
            out['metrics_%s/frobenius_norm' % tag] = np.array(model.frobenius_norm())
    out['meta/k'] = np.array(k); out['meta/sweeps'] = np.array(sweeps); out['meta/tau'] = np.array(0.5)
    out['meta/name'] = np.array('SparseZIGaP')
    return out


def tables(oriana):
    from oriana.utils import digamma, inverse_digamma, sigmoid, logit
    out = {}
    with np.errstate(all='ignore'):
        xs = np.asarray([-2.3, 1.5, 0.45, -0.78, 5.3, -.2, 0., -745., 745., -36., 36., 1e-15])   # test.py:14
        out['sigmoid/x'] = xs
        out['sigmoid/y'] = sigmoid(xs)
        ps = np.asarray([0.45, 0.001, 0.9987, 0.63, 0.745, 0.521, 0.32, 0., 1., 1e-15, 1e-16, 1. - 1e-10, 1e-10])  # test.py:19
        out['logit/x'] = ps
        out['logit/y'] = logit(ps)
        gs = np.asarray([0.54, 6.2, 1.2, 0.3, 7.9, 4.5, 2.1, 1e-15, 1e-8, 1e-3, 1., 1.4616321449683623,
                         10., 1e3, 1e6, 3.7e8])                                                     # test.py:24
        out['digamma/x'] = gs
        out['digamma/y'] = digamma(gs)
        out['digamma/y32'] = digamma(gs.astype(np.float32))
        ys = np.concatenate([digamma(gs), np.asarray([0.54, 6.2, 1.2, 0.3, 7.9, 4.5, 2.1, -2.22, -2.2200001,
                                                     -5., -50., -1e3, -1e15, 0., 13.8, 20.])])
        out['inverse_digamma/x'] = ys
        out['inverse_digamma/y'] = inverse_digamma(ys)
    return out


def main():
    oriana = import_reference()
    np.savez_compressed(os.path.join(HERE, 'tables.npz'), **tables(oriana))
    shapes = {'c1': (200, 80, 5, 0), 'odd': (257, 131, 7, 100)}
    for name in ('GaP', 'ZIGaP', 'SparseGaP', 'SparseZIGaP'):
        for tag, (n, m, k, seed) in shapes.items():
            for uf in (False, True):
                out = run_case(oriana, name, n, m, k, uf, seed + (10 if uf else 0))
                fn = '%s_%s_%s.npz' % (name.lower(), tag, 'nmf' if uf else 'rand')
                np.savez_compressed(os.path.join(HERE, fn), **out)
                print('wrote', fn, 'zeros=%.3f' % (out['X'] == 0).mean())


def xcheck_case(oriana, g):
    """SURVEY 8(a) policy (ii): the patched SparseGaP fixture cross-checked by the reference's UNPATCHED SparseZIGaP class
    with D_hat == 1 (sparse_zigap.py:114-116, 140, 147-148, 155 then reduce to sparse_gap.py:95-97, 119-120, 127-128, 136).
    For each single sweep s_a -> s_b of a sparsegap golden `g`: a SparseZIGaP model on the same X gets the golden's state
    s_a (parameters and expectations), D_hat = 1 (float32), runs its own step(), and the keys the two models share are
    recorded as xcheck/s_b/<key>.  Data only: the golden's inputs and the reference's outputs."""
    from oriana.singlecell import CountMatrix
    X = np.asarray(g['X'])
    k = int(g['meta/k'])
    out = {'X': X, 'meta/k': np.array(k), 'meta/tau': np.array(float(g['meta/tau'])),
           'meta/what': np.array('reference SparseZIGaP.step() with D_hat == 1 from the sparsegap golden states')}
    np.random.seed(12345)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = oriana.models.SparseZIGaP(CountMatrix(X), k=k, use_factors=False, tau=float(g['meta/tau']))
        for a, b in (('s0', 's1'), ('s1', 's2'), ('s2', 's3')):
            for key in ('alpha1', 'alpha2', 'beta1', 'beta2', 'a1', 'a2', 'b1', 'b2', 'pi_s', 'p_s'):
                getattr(model, key)[:] = g['%s/%s' % (a, key)]
            model.U_hat = np.array(g[a + '/U_hat']); model.log_U_hat = np.array(g[a + '/log_U_hat'])
            model.Vprime_hat = np.array(g[a + '/V_hat']); model.log_Vprime_hat = np.array(g[a + '/log_V_hat'])
            model.S_hat = np.array(g[a + '/S_hat'])
            model.D_hat = np.ones(X.shape, dtype=np.float32)
            model.pi_d[:] = 1.0
            model.step()
            tmp = {}
            snapshot(model, b, tmp)
            for kk, v in tmp.items():
                if kk.split('/')[1] not in ('p_d', 'pi_d'):
                    out['xcheck/' + kk] = v
    return out


def main_xcheck():
    oriana = import_reference()
    import glob
    for path in sorted(glob.glob(os.path.join(HERE, 'sparsegap_*_*.npz'))):
        if path.endswith('_xcheck.npz'):
            continue
        g = dict(np.load(path))
        out = xcheck_case(oriana, g)
        worst = 0.0
        for kk, v in out.items():
            if kk.startswith('xcheck/'):
                ref = np.asarray(g[kk[len('xcheck/'):]], dtype=np.float64)
                d = np.abs(np.asarray(v, dtype=np.float64) - ref) / (np.abs(ref) + np.abs(ref).max(axis=0, keepdims=True) + 1e-300)
                worst = max(worst, float(np.nanmax(d)))
        fn = os.path.basename(path)[:-4] + '_xcheck.npz'
        np.savez_compressed(os.path.join(HERE, fn), **out)
        print('wrote', fn, 'largest column-relative difference from the patched-SparseGaP golden: %.3e' % worst)


def main_metrics():
    oriana = import_reference()
    for tag, (n, m, k, seed) in {'c1': (200, 80, 5, 0), 'odd': (257, 131, 7, 100)}.items():
        out = metrics_case(oriana, n, m, k, seed)
        fn = 'metrics_sparsezigap_%s.npz' % tag
        np.savez_compressed(os.path.join(HERE, fn), **out)
        print('wrote', fn, {k: float(v) for k, v in out.items() if k.startswith('metrics_')})


def generator_moments(oriana, n, m, k, z, seeds):
    """Moments of oriana.singlecell.generation.generate_factor_matrices (generation.py:8-86) over several seeds,
    per statistic: [mean over seeds, standard deviation over seeds].  Statistics: block / off-block means of U
    divided by the block's own scale (so that the random choice of alpha in {100, 250} / k drops out), block /
    off-block means of V, mean and variance of the per-gene expression probability estimated from D = (X > 0)
    restricted to cells with Lambda >= 1, the zero fraction of X, the mean of the non-zero counts relative to the
    mean rate."""
    from oriana.singlecell.generation import generate_factor_matrices
    stats = {}

    def put(name, v):
        stats.setdefault(name, []).append(float(v))
    for seed in seeds:
        np.random.seed(seed)
        X, U, V, labels = generate_factor_matrices(n, m, k, zero_inflation_level=z)
        ng = 2
        rows = [np.arange(n)[labels == g] for g in range(ng)]
        kcut = [0, k // ng, k]
        m0 = int(np.round(m * 0.5))
        vcut = [0, m0 // ng, m0]
        # U: in-block entries have scale alpha_g, off-block (1 - theta) * mean(alpha): report ratios that do not
        # depend on the draw of alpha
        inb = np.concatenate([U[np.ix_(rows[g], np.arange(kcut[g], kcut[g + 1]))].ravel() for g in range(ng)])
        offb = np.concatenate([U[np.ix_(rows[g], np.arange(kcut[1 - g], kcut[2 - g]))].ravel() for g in range(ng)])
        put('U_off_over_in_mean', offb.mean() / inb.mean())
        put('U_in_cv', inb.std() / inb.mean())
        vin = np.concatenate([V[vcut[g]:vcut[g + 1], kcut[g]:kcut[g + 1]].ravel() for g in range(ng)])
        voff_rows = np.concatenate([V[vcut[g]:vcut[g + 1], kcut[1 - g]:kcut[2 - g]].ravel() for g in range(ng)])
        vrest = V[m0:].ravel()
        put('V_in_mean', vin.mean())
        put('V_off_mean', np.concatenate([voff_rows, vrest]).mean())
        put('zero_fraction', (X == 0).mean())
        pi_hat = (X > 0).mean(axis=0)
        put('pi_hat_mean', pi_hat.mean())
        put('pi_hat_var', pi_hat.var())
        lam = U @ V.T
        put('nonzero_mean_over_rate', X[X > 0].mean() / lam.mean())
    return {kk: np.asarray([np.mean(v), np.std(v)]) for kk, v in stats.items()}


def main_generator():
    oriana = import_reference()
    out = {'meta/n': 2000, 'meta/m': 400, 'meta/k': 10, 'meta/seeds': 24}
    for z in (0.5, 0.1):
        mom = generator_moments(oriana, 2000, 400, 10, z, range(24))
        for kk, v in mom.items():
            out['z%02d/%s' % (int(round(z * 100)), kk)] = v
            print('z=%.2f %-26s mean %.5f  sd over seeds %.5f' % (z, kk, v[0], v[1]))
    np.savez_compressed(os.path.join(HERE, 'generator_moments.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'metrics':
        main_metrics()          # only the metrics fixtures (the sweep fixtures are left as they are)
    elif len(sys.argv) > 1 and sys.argv[1] == 'xcheck':
        main_xcheck()           # the patched SparseGaP fixtures cross-checked by the unpatched SparseZIGaP class (D_hat == 1)
    elif len(sys.argv) > 1 and sys.argv[1] == 'generator':
        main_generator()        # moments of the reference's synthetic-data generator
    else:
        main()
        main_metrics()
        main_generator()
        main_xcheck()

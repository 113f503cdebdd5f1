# -*- coding: utf-8 -*-
"""Host-side replay of the reference's constructor randomness.

The reference draws every initial value from NumPy's GLOBAL random state, in a fixed order, and
runs scikit-learn's NMF (random_state=None, hence the same global state) in the middle
(reference models/base.py:15-41 and the build_* / define_* methods of each model).  Replaying
the same calls in the same order makes ``np.random.seed(s); Model(X, k)`` start from the same
a1 / b1 as the reference does for that seed (given the same scikit-learn).  Values the first
M-step overwrites (alpha1, beta1, pi_d, pi_s, the gamma(2) placeholders) are drawn and dropped.
"""
import warnings

import numpy as np

__all__ = ['reference_initial_shapes']


def _is_sparse(X):
    import scipy.sparse as sp
    return sp.issparse(X)


def reference_initial_shapes(model_name, X, k, use_factors):
    n, m = X.shape
    zi = model_name in ('ZIGaP', 'SparseZIGaP')
    sparse = model_name in ('SparseGaP', 'SparseZIGaP')
    if model_name != 'GaP':
        np.random.gamma(2., size=k)                 # alpha1   (zigap.py:22, sparse_gap.py:22)
    if sparse:
        np.random.rand(m)                           # pi_s     (sparse_gap.py:27)
    if model_name != 'GaP':
        np.random.gamma(2., size=k)                 # beta1    (zigap.py:27, sparse_gap.py:29)
    if zi:
        np.random.rand(m)                           # pi_d     (zigap.py:32)
    np.random.gamma(2., size=(n, k))                # a1 placeholder (gap.py:37)
    np.random.gamma(2., size=(m, k))                # b1 placeholder (gap.py:42)
    try:
        from sklearn.decomposition import NMF
    except ImportError as e:                        # pragma: no cover
        raise ImportError('scikit-learn is needed to replay the reference initialisation; '
                          'pass init=(a1, b1) instead') from e
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        nmf = NMF(n_components=k)                   # base.py:38
        # base.py:39 (a SciPy sparse X goes to scikit-learn as it is: same factorisation, no dense copy)
        W = nmf.fit_transform(X if _is_sparse(X) else np.asarray(X))
        H = nmf.components_.T                       # base.py:40
    if use_factors:
        a1, b1 = W, H                               # gap.py:49-50, 59-60
    else:
        a1 = np.random.gamma(1., size=(n, k))       # gap.py:52
        b1 = np.random.gamma(1., size=(m, k))       # gap.py:62
    return (np.ascontiguousarray(a1, dtype=np.float64), np.ascontiguousarray(b1, dtype=np.float64),
            (np.array(W), np.array(H)))

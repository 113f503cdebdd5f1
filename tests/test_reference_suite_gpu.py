# -*- coding: utf-8 -*-
"""Known-answer vectors for the hot-path pieces that the reference's own unit tests pin (reference
test/test.py:13-41, 60-79: sigmoid / logit, digamma / inverse_digamma, Bernoulli.mean, Gamma.mean / meanlog with a
tile + transpose relation), evaluated through the HIP-backed mirrors (oriana_amd.utils, oriana_amd.nodes).

The table below is DATA: (operation under test, input vector, expected vector, decimals).  The inputs are the
points the reference checks; the expected values are what its assertions imply (the input itself for the four
round trips and for the identity-shaped means, the tiled array for the 'n,m,k ~ d,s,d' relation) or were
captured from the reference (tests/golden/tables.npz, written by tests/golden/make_golden.py).
test_multinomial_mean / test_node_forward of the reference exercise nodes that step() never calls: out of scope."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

X_REAL = np.asarray([-2.3, 1.5, 0.45, -0.78, 5.3, -.2, 0.])                   # test/test.py:14
P_UNIT = np.asarray([0.45, 0.001, 0.9987, 0.63, 0.745, 0.521, 0.32])          # test/test.py:19
X_POS = np.asarray([0.54, 6.2, 1.2, 0.3, 7.9, 4.5, 2.1])                      # test/test.py:24, 30
P_22 = np.asarray([[0.02, 0.34], [0.62, 0.79]])                               # test/test.py:36
A_22 = np.asarray([[2.1, 1.8], [0.7, 2.3]])                                   # test/test.py:61
A_TILED = np.asarray([[[2.1, 1.8], [2.1, 1.8]], [[0.7, 2.3], [0.7, 2.3]]])    # 'n,m,k ~ d,s,d': m is a sample axis
# digamma at the four shape values (scipy.special.digamma, float64), tiled the same way
PSI = {2.1: 0.48533596867983236, 1.8: 0.28499143329386156, 0.7: -1.2200235536979347, 2.3: 0.6000398803639695}
PSI_TILED = np.vectorize(PSI.get)(A_TILED)


def _u():
    from oriana_amd import utils
    return utils


def _node(kind, *params, relation):
    from oriana_amd import Dimensions, Parameter, nodes
    dims = Dimensions({'n': 2, 'm': 2, 'k': 2})
    return getattr(nodes, kind)(*[Parameter(p) for p in params], dims(relation))


VECTORS = [
    # id, operation, input, expected, decimals (numpy.testing.assert_almost_equal semantics, as the reference uses)
    ('logit_of_sigmoid', lambda x: _u().logit(_u().sigmoid(x)), X_REAL, X_REAL, 7),
    ('sigmoid_of_logit', lambda p: _u().sigmoid(_u().logit(p)), P_UNIT, P_UNIT, 7),
    ('inverse_digamma_of_digamma', lambda x: _u().inverse_digamma(_u().digamma(x)), X_POS, X_POS, 7),
    ('digamma_of_inverse_digamma', lambda x: _u().digamma(_u().inverse_digamma(x)), X_POS, X_POS, 7),
    ('bernoulli_mean_dd', lambda p: _node('Bernoulli', p, relation='m,k ~ d,d').mean(), P_22, P_22, 7),
    ('gamma_mean_tile_transpose', lambda a: _node('Gamma', a, np.ones((2, 2)), relation='n,m,k ~ d,s,d').mean(), A_22, A_TILED, 7),
    ('gamma_meanlog_tile_transpose', lambda a: _node('Gamma', a, np.ones((2, 2)), relation='n,m,k ~ d,s,d').meanlog(), A_22, PSI_TILED, 6),
]


@pytest.mark.parametrize('name,op,x,expected,decimals', VECTORS, ids=[v[0] for v in VECTORS])
def test_known_answers(name, op, x, expected, decimals):
    got = np.asarray(op(x))
    assert got.shape == expected.shape
    np.testing.assert_almost_equal(got, expected, decimal=decimals)


TABLES = [
    # function, table key, rtol, atol: golden tables captured from the reference, incl. extreme arguments
    ('sigmoid', 'sigmoid', 1e-14, 0.0),
    ('logit', 'logit', 1e-13, 0.0),
    ('digamma', 'digamma', 1e-12, 1e-15),
    ('inverse_digamma', 'inverse_digamma', 1e-9, 0.0),
]


@pytest.mark.parametrize('fn,key,rtol,atol', TABLES, ids=[t[0] for t in TABLES])
def test_special_function_tables(golden_dir, fn, key, rtol, atol):
    t = np.load(os.path.join(golden_dir, 'tables.npz'))
    with np.errstate(all='ignore'):
        got = getattr(_u(), fn)(t[key + '/x'])
    ref = t[key + '/y']
    ok = np.isfinite(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=rtol, atol=atol)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(got[~ok & ~np.isnan(ref)], ref[~ok & ~np.isnan(ref)])        # infinities in place


def test_meanlog_is_float32_of_float64_digamma(golden_dir):
    """Gamma.meanlog casts its parameters to float32 first and SciPy's float32 digamma is the float64 one
    rounded: the float32 column of the golden table, to one float32 ulp."""
    from oriana_amd import Dimensions, Parameter, nodes
    t = np.load(os.path.join(golden_dir, 'tables.npz'))
    x = t['digamma/x']
    keep = (x > 0) & np.isfinite(t['digamma/y32'])
    a = x[keep].reshape(1, -1)
    dims = Dimensions({'n': 1, 'k': a.shape[1]})
    got = nodes.Gamma(Parameter(a), Parameter(np.ones_like(a)), dims('n,k ~ d,d')).meanlog()
    assert got.dtype == np.float32
    ref = t['digamma/y32'][keep].reshape(1, -1) - np.float32(0.0)
    np.testing.assert_allclose(got, ref, rtol=2e-7, atol=1e-12)          # (the table holds the zero of digamma: value -1.2e-8)

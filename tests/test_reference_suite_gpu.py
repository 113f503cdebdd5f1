# -*- coding: utf-8 -*-
"""The reference's own unit tests for the hot-path pieces (reference test/test.py:13-41, 60-79),
re-run against the HIP-backed mirrors: same inputs, same assertions.  (test_multinomial_mean and
test_node_forward exercise nodes that are never called from step(): out of scope.)"""
import numpy as np
import pytest
from numpy.testing import assert_almost_equal

pytestmark = pytest.mark.gpu


def test_sigmoid():
    from oriana_amd.utils import sigmoid, logit
    x = np.asarray([-2.3, 1.5, 0.45, -0.78, 5.3, -.2, 0.])
    assert_almost_equal(logit(sigmoid(x)), x)


def test_logit():
    from oriana_amd.utils import sigmoid, logit
    x = np.asarray([0.45, 0.001, 0.9987, 0.63, 0.745, 0.521, 0.32])
    assert_almost_equal(sigmoid(logit(x)), x)


def test_digamma():
    from oriana_amd.utils import digamma, inverse_digamma
    x = np.asarray([0.54, 6.2, 1.2, 0.3, 7.9, 4.5, 2.1])
    y = inverse_digamma(digamma(x))
    assert_almost_equal(x, y)


def test_digamma_inverse():
    from oriana_amd.utils import digamma, inverse_digamma
    x = np.asarray([0.54, 6.2, 1.2, 0.3, 7.9, 4.5, 2.1])
    y = digamma(inverse_digamma(x))
    assert_almost_equal(x, y)


def test_bernoulli_mean():
    from oriana_amd import Dimensions, Parameter
    from oriana_amd.nodes import Bernoulli
    p = Parameter([[0.02, 0.34], [0.62, 0.79]])
    dims = Dimensions({'n': 2, 'm': 2, 'k': 2})
    bern = Bernoulli(p, dims('m,k ~ d,d'))
    x = np.asarray([[0.02, 0.34], [0.62, 0.79]])
    y = bern.mean()
    assert_almost_equal(x, y)


def test_gamma_mean():
    from oriana_amd import Dimensions, Parameter
    from oriana_amd.nodes import Gamma
    alpha1 = Parameter([[2.1, 1.8], [0.7, 2.3]])
    alpha2 = Parameter(np.ones((2, 2)))
    dims = Dimensions({'n': 2, 'm': 2, 'k': 2})
    gamma = Gamma(alpha1, alpha2, dims('n,m,k ~ d,s,d'))
    x = gamma.mean()
    y = np.asarray([[[2.1, 1.8], [2.1, 1.8]],
                    [[0.7, 2.3], [0.7, 2.3]]])
    assert_almost_equal(x, y)


def test_gamma_mean_log():
    from oriana_amd import Dimensions, Parameter
    from oriana_amd.nodes import Gamma
    from oriana_amd.utils import digamma
    alpha1 = Parameter([[2.1, 1.8], [0.7, 2.3]])
    alpha2 = Parameter(np.ones((2, 2)))
    dims = Dimensions({'n': 2, 'm': 2, 'k': 2})
    gamma = Gamma(alpha1, alpha2, dims('n,m,k ~ d,s,d'))
    x = gamma.meanlog()
    y = digamma(np.asarray([[[2.1, 1.8], [2.1, 1.8]],
                            [[0.7, 2.3], [0.7, 2.3]]]))
    assert_almost_equal(x, y)


def test_special_function_tables(golden_dir):
    """The same functions on the golden tables captured from the reference (incl. extreme arguments)."""
    import os
    from oriana_amd import utils
    t = np.load(os.path.join(golden_dir, 'tables.npz'))
    with np.errstate(all='ignore'):
        np.testing.assert_allclose(utils.sigmoid(t['sigmoid/x']), t['sigmoid/y'], rtol=1e-14, atol=0)
        np.testing.assert_allclose(utils.logit(t['logit/x']), t['logit/y'], rtol=1e-13, atol=0)
        np.testing.assert_allclose(utils.digamma(t['digamma/x']), t['digamma/y'], rtol=1e-12, atol=1e-15)
        got = utils.inverse_digamma(t['inverse_digamma/x']); ref = t['inverse_digamma/y']
        ok = np.isfinite(ref)
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-9)
        assert np.array_equal(np.isnan(got), np.isnan(ref))

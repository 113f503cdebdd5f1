# -*- coding: utf-8 -*-
"""The two distribution nodes whose expectations are on the hot path, with the reference's
constructor and method names: ``Gamma(alpha, beta, rel).mean() / .meanlog()``
(oriana/nodes/probabilistic/gamma.py:37-61) and ``Bernoulli(pi, rel).mean()``
(nodes/probabilistic/bernoulli.py:41-48).

As in the reference (``ProbabilisticNode.updates_buffer``, nodes/base.py:123-143) a call gathers
the parent parameters, evaluates the expectation per distribution, tiles it over the sample axes,
reshapes it through the ``DimRelation`` into the node's shape, overwrites the node buffer with it
and returns it.  The per-distribution arithmetic runs in the HIP kernel ``oriana_gamma_update``
(float32 cast of the parameters before digamma / log, as gamma.py:56-57).  Sampling and
log-densities are out of scope (never called from ``step()``).
"""
import numpy as np
import torch

from ._lib import call, ptr, stream_ptr
from .parameters import Parameter

__all__ = ['Gamma', 'Bernoulli']


def _as_tensor(param):
    if isinstance(param, Parameter):
        return param.tensor
    if isinstance(param, torch.Tensor):
        return param
    return Parameter(param).tensor


class _Node:
    def __init__(self, rel, name=''):
        self.rel = rel
        self.name = name
        self.shape = rel.shape
        self.n_samples_per_distrib = rel.n_samples_per_distrib
        self.n_distribs = rel.n_distribs
        self.n_components = rel.n_components
        self._buffer = None

    @property
    def buffer(self):
        return self._buffer

    def asarray(self):
        return self._buffer.cpu().numpy()

    def __getitem__(self, key):
        return self._buffer[key].cpu().numpy()

    def _publish(self, per_distrib):
        """(n_distribs,) device tensor -> tiled (n_samples, n_distribs, 1) -> node shape; stored and returned."""
        out = per_distrib.reshape(1, -1).repeat(self.n_samples_per_distrib, 1).unsqueeze(-1)
        out = self.rel.reshape_func(out).contiguous()
        assert tuple(out.shape) == tuple(self.shape)
        self._buffer = out.to(torch.float64)         # the reference's node buffers are float64
        return out.cpu().numpy()

    def __repr__(self):
        return 'Variable %s of shape %s' % (self.name, str(self.shape))


class Gamma(_Node):
    """Gamma(shape alpha, rate beta) node: E[x] = alpha / beta, E[log x] = digamma(alpha) - log(beta)."""

    def __init__(self, alpha, beta, rel, **kwargs):
        _Node.__init__(self, rel, **kwargs)
        self.parents = (alpha, beta)

    def _expectations(self):
        a = _as_tensor(self.parents[0]).to(device='cuda', dtype=torch.float64).reshape(-1).contiguous().clone()
        b = _as_tensor(self.parents[1]).to(device='cuda', dtype=torch.float64).reshape(-1).contiguous().clone()
        assert a.numel() == self.n_distribs and b.numel() == self.n_distribs
        E = torch.empty_like(a)
        Elog = torch.empty(a.numel(), dtype=torch.float32, device='cuda')
        # Z = NULL: parameters are taken as they are, only the expectations are produced
        call('oriana_gamma_update', ptr(a), ptr(b), ptr(E), ptr(Elog), None, None, None, None, None, None, None, None,
             None, a.numel(), 1, stream_ptr())
        return E, Elog

    def mean(self):
        return self._publish(self._expectations()[0])

    def meanlog(self):
        return self._publish(self._expectations()[1])


class Bernoulli(_Node):
    """Bernoulli(pi) node: E[x] = float32(pi) (bernoulli.py:45)."""

    def __init__(self, pi, rel, **kwargs):
        _Node.__init__(self, rel, **kwargs)
        self.parents = (pi,)

    def mean(self):
        p = _as_tensor(self.parents[0]).to(device='cuda').reshape(-1).to(torch.float32)
        return self._publish(p)

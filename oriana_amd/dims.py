# -*- coding: utf-8 -*-
"""Shape algebra of the node graph: ``Dimensions('n,k ~ s,d')`` -> ``DimRelation``.

Mirrors the behaviour of the reference's ``oriana/dims.py`` (``Dimensions.__call__`` :77-151,
``DimRelation`` :11-60): the left side names the axes of a node buffer, the right side tags each
axis as sample ('s'), distribution ('d') or component ('c').  The relation maps between the buffer
shape and the canonical (n_samples_per_distrib, n_distribs, n_components) triple.  Pure integer /
permutation bookkeeping -- results are exact.  Works on NumPy arrays and torch tensors alike.
"""
from functools import reduce
from operator import mul

from .exceptions import IncompatibleShapeException

__all__ = ['DimRelation', 'Dimensions']

_TAGS = ('s', 'd', 'c')


def _prod(xs):
    return reduce(mul, xs, 1)


def _permute(data, axes):
    if hasattr(data, 'permute'):          # torch
        return data.permute(*axes)
    return data.transpose(axes)


class DimRelation:
    """Mapping buffer shape <-> (samples, distributions, components)."""

    def __init__(self, shape, n_samples_per_distrib, n_distribs, n_components, reshape_func, inv_reshape_func):
        self.shape = shape
        self.n_samples_per_distrib = n_samples_per_distrib
        self.n_distribs = n_distribs
        self.n_components = n_components
        self.reshape_func = reshape_func
        self.inv_reshape_func = inv_reshape_func

    @property
    def canonical_shape(self):
        return (self.n_samples_per_distrib, self.n_distribs, self.n_components)

    def is_identity(self):
        """True when the canonical array, flattened, is the buffer, flattened (no permutation,
        no replication across samples) -- the case of every 'd,d' relation used by the models."""
        return self._identity

    def __repr__(self):
        return 'Dimension mapping %s <-> %s' % (str(self.shape), str(self.canonical_shape))


class Dimensions:
    """Named dimensions; calling with a relation string instantiates a DimRelation."""

    def __init__(self, dims):
        self.dims = dims

    def __call__(self, rel):
        try:
            left, right = rel.split('~')
        except ValueError:
            raise IncompatibleShapeException('Relation "%s" format is not correct.' % rel)
        names = [tok.strip() for tok in left.strip().split(',')]
        tags = [tok.strip() for tok in right.strip().split(',')]
        if len(names) != len(tags):
            raise IncompatibleShapeException('Relation "%s" format is not correct.' % rel)

        shape = tuple(self.dims[name] for name in names)
        # axes grouped by tag, in the order s, d, c (axes with an unknown tag are ignored, as in
        # the reference)
        groups = {tag: [ax for ax, t in enumerate(tags) if t == tag] for tag in _TAGS}
        order = groups['s'] + groups['d'] + groups['c']
        sizes = {tag: [shape[ax] for ax in groups[tag]] for tag in _TAGS}
        canonical = tuple(_prod(sizes[tag]) for tag in _TAGS)
        grouped_shape = tuple(sizes['s'] + sizes['d'] + sizes['c'])
        inverse_order = [order.index(ax) for ax in range(len(order))]

        def reshape(data):
            assert tuple(data.shape) == canonical
            return _permute(data.reshape(*grouped_shape), order)

        def inv_reshape(data):
            assert tuple(data.shape) == shape
            return _permute(data, inverse_order).reshape(*canonical)

        rel_obj = DimRelation(shape, canonical[0], canonical[1], canonical[2], reshape, inv_reshape)
        rel_obj._identity = (order == sorted(order)) and canonical[0] == 1
        return rel_obj

    def __setitem__(self, key, value):
        # The reference's __setitem__ (dims.py:160) evaluates ``self.dims[key, value]`` and stores
        # nothing; here the evident intent is implemented.
        self.dims[key] = value

    def __getitem__(self, key):
        return self.dims[key]

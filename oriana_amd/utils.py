# -*- coding: utf-8 -*-
"""Special functions of the hot path, with the reference's names and semantics
(oriana/utils.py:9-15, 31-51), evaluated by the float64 HIP kernels of csrc/updates.hip.

Inputs: NumPy arrays / scalars (returned as NumPy, like the reference) or torch tensors (returned
as device tensors).  There is no CPU fallback: without a GPU these raise.
"""
import numpy as np
import torch

from ._lib import call, ptr, stream_ptr, OrianaHipError

__all__ = ['digamma', 'digamma_prime', 'inverse_digamma', 'sigmoid', 'logit']


def _map(entry, x):
    if not torch.cuda.is_available():
        raise OrianaHipError('oriana_amd.utils needs a ROCm GPU (no CPU fallback)')
    as_tensor = isinstance(x, torch.Tensor)
    t = x if as_tensor else torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
    shape = tuple(t.shape)
    d = t.to(device='cuda', dtype=torch.float64).contiguous().view(-1)
    y = torch.empty_like(d)
    call(entry, ptr(y), ptr(d), d.numel(), stream_ptr())
    y = y.view(shape)
    if as_tensor:
        return y
    out = y.cpu().numpy()
    return out if out.ndim else out[()]


def digamma(x):
    """scipy.special.digamma (utils.py:31-32)."""
    return _map('oriana_digamma_f64', x)


def digamma_prime(x):
    """scipy.special.polygamma(1, x) (utils.py:35-36)."""
    return _map('oriana_trigamma_f64', x)


def inverse_digamma(y):
    """Minka's initialisation + 5 Newton steps (utils.py:39-51)."""
    return _map('oriana_inverse_digamma_f64', y)


def sigmoid(x):
    """1 / (1 + exp(-x)) (utils.py:14-15)."""
    return _map('oriana_sigmoid_f64', x)


def logit(x):
    """log(x / (1 - x)) after clipping x to [1e-15, 1 - 1e-15] (utils.py:9-11)."""
    return _map('oriana_logit_f64', x)

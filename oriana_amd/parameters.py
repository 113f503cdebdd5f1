# -*- coding: utf-8 -*-
"""``Parameter``: the float64 buffer that holds every variational / prior parameter.

Same surface as the reference's ``oriana/parameters.py:8-32`` (``asarray()``, ``[]`` get / set,
``.shape``, ``.buffer`` with a setter); the buffer is a float64 torch tensor resident on the
model's device instead of a host ndarray.  ``p[key]`` and ``p.asarray()`` return host NumPy
copies (what the reference returns); ``p.buffer`` / ``p.tensor`` is the device buffer itself,
which is what the kernels update in place.
"""
import numpy as np
import torch

__all__ = ['Parameter']


def _default_device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


class Parameter:

    def __init__(self, data, device=None):
        dev = torch.device(device) if device is not None else None
        if isinstance(data, torch.Tensor):
            self._buffer = data.to(dtype=torch.float64, device=dev if dev is not None else data.device)
        else:
            host = np.asarray(data, dtype=np.float64)
            self._buffer = torch.from_numpy(np.ascontiguousarray(host)).to(dev if dev is not None else _default_device())

    def asarray(self):
        return self._buffer.detach().cpu().numpy()

    def __getitem__(self, key):
        out = self._buffer[key]
        return out.detach().cpu().numpy() if out.dim() else out.item()

    def __setitem__(self, key, value):
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value))
        if isinstance(value, torch.Tensor):
            value = value.to(device=self._buffer.device, dtype=torch.float64)
        if isinstance(key, np.ndarray):
            key = torch.from_numpy(key).to(self._buffer.device)
        self._buffer[key] = value

    @property
    def shape(self):
        return tuple(self._buffer.shape)

    @property
    def tensor(self):
        return self._buffer

    @property
    def buffer(self):
        return self._buffer

    @buffer.setter
    def buffer(self, data):
        if isinstance(data, torch.Tensor):
            self._buffer = data.to(dtype=torch.float64, device=self._buffer.device)
        else:
            self._buffer = torch.from_numpy(np.ascontiguousarray(np.asarray(data, dtype=np.float64))).to(self._buffer.device)

    def __repr__(self):
        return 'Parameter(shape=%s, device=%s)' % (self.shape, self._buffer.device)

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

__all__ = ['Parameter', 'LazyParameter']


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
        dev = self._buffer.device
        if isinstance(data, torch.Tensor):
            self._buffer = data.to(dtype=torch.float64, device=dev)
        else:
            self._buffer = torch.from_numpy(np.ascontiguousarray(np.asarray(data, dtype=np.float64))).to(dev)

    def __repr__(self):
        return 'Parameter(shape=%s, device=%s)' % (self.shape, self._buffer.device)


class LazyParameter(Parameter):
    """A Parameter whose float64 buffer is produced on first access by ``source()`` and dropped
    again by ``defer(source)``.  Used for the (n, p) dropout posterior ``p_d`` of the ZI models
    (reference zigap.py:41-43): the sweep itself only needs ``D_hat = float32(p_d)`` and the column
    sums of ``p_d``, so the 8 n p bytes of the float64 matrix exist only while somebody looks at
    them (``model.p_d[:]``, ``state()``) or after they were written from outside."""

    def __init__(self, shape, device, source):
        self._shape = tuple(int(v) for v in shape)
        self._device = torch.device(device)
        self._buf = None
        self._source = source

    # the base class reads and writes self._buffer everywhere
    @property
    def _buffer(self):
        if self._buf is None:
            t = self._source()
            assert t.dtype == torch.float64 and tuple(t.shape) == self._shape
            self._buf = t
        return self._buf

    @_buffer.setter
    def _buffer(self, t):
        self._buf = t

    def defer(self, source):
        """Forget the materialised buffer; the next access evaluates ``source()``."""
        self._buf = None
        self._source = source

    @property
    def materialised(self):
        return self._buf is not None

    @property
    def buffer(self):
        return self._buffer

    @buffer.setter
    def buffer(self, data):
        if isinstance(data, torch.Tensor):
            self._buf = data.to(dtype=torch.float64, device=self._device)
        else:
            self._buf = torch.from_numpy(np.ascontiguousarray(np.asarray(data, dtype=np.float64))).to(self._device)

    @property
    def shape(self):
        return self._shape

    def __repr__(self):
        return 'LazyParameter(shape=%s, device=%s, materialised=%s)' % (self._shape, self._device, self.materialised)

# -*- coding: utf-8 -*-
"""``CountMatrix``: the container the models are constructed from (cells x genes).

Same surface as the reference's ``oriana/singlecell/cmatrix.py:12-115`` -- ``as_array()``,
``as_sparse_matrix(mode)``, ``from_csv(...)``, ``T``, ``shape``, ``col_names``, ``row_names``,
``[]`` get / set, ``filter_rows`` -- because the models only ever ask for ``.shape`` and
``.as_array()`` (reference ``models/base.py:22-23``, ``gap.py:31``).  Two additions for data that is
sparse to begin with (real single-cell matrices are > 90 % zeros): the
constructor also takes a SciPy sparse matrix, and ``to_tiles(device)`` packs the counts into the
device-resident tile layout chunk by chunk without ever forming the dense matrix on the host.
"""
import numpy as np
import pandas as pd
import scipy.sparse as sp

from ..exceptions import DatatypeException

__all__ = ['CountMatrix']


class CountMatrix:

    def __init__(self, data):
        self._frame = None
        self._sparse = None
        if isinstance(data, pd.DataFrame):
            self._frame = data
        elif isinstance(data, np.ndarray):
            self._frame = pd.DataFrame(data=data)
        elif sp.issparse(data):
            self._sparse = data.tocsr()
        else:
            raise DatatypeException('Incompatible type %s' % type(data))

    # ---- what the models read -----------------------------------------------------------------------
    @property
    def is_sparse(self):
        return self._sparse is not None

    @property
    def shape(self):
        return self._sparse.shape if self.is_sparse else self._frame.shape

    def as_array(self):
        """Dense ndarray of the counts (cells x genes)."""
        if self.is_sparse:
            return np.asarray(self._sparse.todense())
        return self._frame.values

    def as_sparse_matrix(self, mode='csc'):
        """SciPy sparse copy.  The reference returns CSC for every ``mode`` (cmatrix.py:49-53); 'csr'
        gives CSR here."""
        src = self._sparse if self.is_sparse else self.as_array()
        return sp.csr_matrix(src) if mode == 'csr' else sp.csc_matrix(src)

    def to_tiles(self, device='cuda', chunk_rows=8192, reduce_fn=None):
        """Pack into the HIP engine's resident layout (engine.CountTiles), streaming row chunks."""
        from .. import engine
        if self.is_sparse:
            return engine.CountTiles.from_scipy(self._sparse, device, chunk_rows=chunk_rows, reduce_fn=reduce_fn)
        return engine.CountTiles.from_dense(self.as_array(), device, reduce_fn=reduce_fn)

    # ---- construction ----------------------------------------------------------------------------------
    @staticmethod
    def from_csv(filepath, delimiter=',', has_col_names=True, has_row_names=True):
        """Cells in rows, genes in columns; optional header row / first column of names."""
        frame = pd.read_csv(filepath, sep=delimiter, header=0 if has_col_names else None,
                            index_col=0 if has_row_names else False, skip_blank_lines=True)
        return CountMatrix(frame)

    # ---- labelled access (dense, DataFrame-backed matrices) ------------------------------------
    def _need_frame(self):
        if self._frame is None:
            self._frame = pd.DataFrame(data=self.as_array())
            self._sparse = None
        return self._frame

    @property
    def T(self):
        if self.is_sparse:
            return CountMatrix(self._sparse.T)
        return CountMatrix(self._frame.transpose(copy=False))

    @property
    def col_names(self):
        return self._need_frame().columns.values

    @property
    def row_names(self):
        return np.asarray(self._need_frame().index)

    def __getitem__(self, key):
        return self._need_frame()[key]

    def __setitem__(self, key, value):
        self._need_frame()[key] = value

    def filter_rows(self, rows, inplace=True):
        kept = self._need_frame().loc[rows]
        if not inplace:
            return CountMatrix(pd.DataFrame(kept))
        self._frame = kept
        return self

    def __repr__(self):
        if self.is_sparse:
            return 'CountMatrix(sparse, shape=%s, nnz=%d)' % (self.shape, self._sparse.nnz)
        return repr(self._frame)

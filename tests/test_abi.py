# -*- coding: utf-8 -*-
"""The C-ABI library loads and exports every function include/oriana_hip.h declares (no compute:
runs without a GPU).  Also: the ctypes binding table covers exactly the declared functions."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'oriana_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(oriana_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    return ctypes.CDLL(g.build())


def test_header_declares_functions():
    names = _declared()
    assert 'oriana_row_pass' in names and 'oriana_col_pass' in names and len(names) >= 15


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    from oriana_amd import _lib
    assert sorted(_lib.declared_symbols()) == _declared()


def test_kpad_and_version(lib):
    lib.oriana_kpad.restype = ctypes.c_int64
    lib.oriana_kpad.argtypes = [ctypes.c_int64]
    assert lib.oriana_kpad(5) == 16 and lib.oriana_kpad(20) == 20 and lib.oriana_kpad(100) == 100 and lib.oriana_kpad(101) == 112
    assert lib.oriana_kpad(256) == 256 and lib.oriana_kpad(0) == 0 and lib.oriana_kpad(257) == 0
    lib.oriana_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.oriana_version()


def test_argument_errors_without_gpu(lib):
    """Argument validation happens before any HIP call: usable on a CPU-only host."""
    lib.oriana_factor_prep.restype = ctypes.c_int
    lib.oriana_factor_prep.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 2 + [ctypes.c_void_p]
    assert lib.oriana_factor_prep(None, None, None, None, None, -1, 3, None) == -1       # ORIANA_EINVAL
    assert lib.oriana_factor_prep(None, None, None, None, None, 4, 1000, None) == -2     # ORIANA_EKRANGE
    assert lib.oriana_factor_prep(None, None, None, None, None, 0, 3, None) == 0         # empty input is fine


def test_missing_library_fails_loudly(monkeypatch):
    from oriana_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/liboriana_hip.so')
    with pytest.raises(_lib.OrianaHipError):
        _lib.load()

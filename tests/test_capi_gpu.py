# -*- coding: utf-8 -*-
"""The C-ABI boundary from plain C++ / HIP (examples/capi_zq_gap.cpp): no Python, no PyTorch in the
process that calls oriana_zq_gap_f32.  Built with hipcc on the GPU box, run as a child process, checked
against the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import err_colrel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if c and os.path.exists(c):
            return c
    return None


def _build_example(tmp_path):
    import oriana_amd._build as B
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip('hipcc not available')
    B.build()
    csrc = os.path.join(ROOT, 'oriana_amd', 'csrc')
    exe = str(tmp_path / 'capi_zq_gap')
    cmd = [hipcc, '--offload-arch=gfx950', '-O2', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'capi_zq_gap.cpp'), '-L' + csrc, '-loriana_hip', '-Wl,-rpath,' + csrc, '-o', exe]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return exe


@pytest.mark.parametrize('K,dense_density', [(20, 0.0), (20, 0.12), (100, 0.0), (100, 0.12)])
def test_capi_resident_handle_from_cpp(tmp_path, K, dense_density):
    """[r5] Create once, call twenty times, from a process with no Python and no PyTorch: oriana_counts_create_dense_f32 packs X
    and plans the passes in C host code (gene order, dense genes, column work list, row split), oriana_zq_gap_resident is the
    nest of gap.py:67-80 on that layout -- sliced, and hybrid (the densest genes on the matrix cores).  First and last call
    against the oracle."""
    from oracle import cavi_oracle as co
    exe = _build_example(tmp_path)
    rng = np.random.default_rng(5 + K)
    n, m = 700, 530
    dens = rng.beta(1.0, 6.0, size=m)
    X = (rng.poisson(3.0, size=(n, m)) * (rng.random((n, m)) < dens[None, :])).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32)
    lv = rng.normal(size=(m, K)).astype(np.float32)
    fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(fin, 'wb') as f:
        np.array([n, m, K], dtype=np.int64).tofile(f)
        lu.tofile(f); lv.tofile(f); X.tofile(f)
    out = subprocess.run([exe, fin, fout, '20', str(dense_density)], check=True, capture_output=True, text=True)
    assert 'resident' in out.stdout and 'calls=20' in out.stdout, out.stdout
    fields = dict(kv.split('=') for kv in out.stdout.split() if '=' in kv)
    assert int(fields['nnz']) == int(np.count_nonzero(X))
    assert (int(fields['dense_genes']) > 0) == (dense_density > 0) and int(fields['dense_genes']) % 32 == 0
    got = np.fromfile(fout, dtype=np.float32)
    assert got.size == 2 * (n + m) * K
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    co.zq_gap(rZi, rZj, lu, lv, X)
    for call in range(2):
        blk = got[call * (n + m) * K:(call + 1) * (n + m) * K]
        Zi, Zj = blk[:n * K].reshape(n, K), blk[n * K:].reshape(m, K)
        assert err_colrel(Zi, rZi) < 1e-5 and err_colrel(Zj, rZj) < 1e-5, call
        assert abs(float(Zi.sum(dtype=np.float64)) - float(X.sum(dtype=np.float64))) < 1e-5 * float(X.sum())


def test_capi_from_cpp(tmp_path):
    from oracle import cavi_oracle as co
    exe = _build_example(tmp_path)
    rng = np.random.default_rng(21)
    n, m, K = 300, 270, 20
    X = (rng.poisson(3.0, size=(n, m)) * (rng.random((n, m)) < 0.2)).astype(np.float32)
    lu = rng.normal(size=(n, K)).astype(np.float32)
    lv = rng.normal(size=(m, K)).astype(np.float32)
    fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(fin, 'wb') as f:
        np.array([n, m, K], dtype=np.int64).tofile(f)
        lu.tofile(f); lv.tofile(f); X.tofile(f)
    out = subprocess.run([exe, fin, fout], check=True, capture_output=True, text=True)
    assert 'oriana_hip' in out.stdout
    got = np.fromfile(fout, dtype=np.float32)
    Zi, Zj = got[:n * K].reshape(n, K), got[n * K:].reshape(m, K)
    rZi = np.empty((n, K), np.float32); rZj = np.empty((m, K), np.float32)
    co.zq_gap(rZi, rZj, lu, lv, X)
    assert err_colrel(Zi, rZi) < 1e-5 and err_colrel(Zj, rZj) < 1e-5

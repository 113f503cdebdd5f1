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


def test_capi_from_cpp(tmp_path):
    from oracle import cavi_oracle as co
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

# -*- coding: utf-8 -*-
"""The float32-equivalence of the bf16 x 3 matrix-core evaluation, re-measured on the GPU box every round.

The dense-gene kernels (csrc/dense_pass.hip) and the ZI sweep (csrc/dense_f32.hip, csrc/dense_zi.hip) evaluate float32
products as six bf16 cross products of exact three-way splits with float32 accumulation and keep no chain beyond 256
terms on the matrix core.  tools/ubench/mfma_bf16x3.hip measures the error of such a chain against float64, next to a
float32 FMA chain of the same terms (round 2: profiles/r02_ubench_mfma_bf16x3.txt -- 64 terms: mean -4.6e-8, rms 1.0e-7,
max 3.5e-7; 512 terms: mean -8.9e-8, rms 2.9e-7, max 9.1e-7).  The bounds below are <= 2 x those figures."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# terms: (|mean signed relative error|, rms, max) bounds; and rms <= RATIO x the float32 FMA chain's rms
BOUNDS = {64: (9e-8, 1.6e-7, 7e-7), 256: (1.4e-7, 3.2e-7, 1.4e-6), 512: (1.8e-7, 4.5e-7, 1.8e-6)}
RATIO = 1.25


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if c and os.path.exists(c):
            return c
    return None


def test_bf16x3_chain_error_is_the_float32_chains(tmp_path):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip('hipcc not available')
    exe = str(tmp_path / 'mfma_bf16x3')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O2', '-o', exe, os.path.join(ROOT, 'tools', 'ubench', 'mfma_bf16x3.hip')],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([exe, 'errors'], check=True, capture_output=True, text=True, timeout=300).stdout
    pat = re.compile(r'\s*(\d+) terms: .*mean signed rel err ([-+.\de]+)\s+rms ([-+.\de]+)\s+max ([-+.\de]+)\s+\| float32 fma chain rms ([-+.\de]+)')
    seen = {}
    for line in out.splitlines():
        mt = pat.match(line)
        if mt:
            seen[int(mt.group(1))] = tuple(float(mt.group(i)) for i in range(2, 6))
    assert set(seen) == set(BOUNDS), out
    for terms, (mean, rms, mx, rms32) in seen.items():
        bm, br, bx = BOUNDS[terms]
        assert abs(mean) <= bm, (terms, 'drift', mean)
        assert rms <= br and mx <= bx, (terms, rms, mx)
        assert rms <= RATIO * rms32, (terms, 'rms against the float32 FMA chain', rms, rms32)

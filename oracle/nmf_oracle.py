# -*- coding: utf-8 -*-
"""oracle/nmf_oracle.py -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).

NumPy restatement of the warm start of the factor models.  The reference calls scikit-learn's
``NMF(n_components=k).fit_transform(X)`` (oriana/models/base.py:38-40) -- a third-party routine absent from
/root/reference and unpinned there (``requirements.txt`` names no scikit-learn version; SURVEY.md 8c treats its
output as an INPUT fixture).  What the GPU path offers for data that never visits the host
(oriana_amd/models/deviceinit.py: device_nmf) is the published multiplicative-update algorithm for the Frobenius
loss (Lee & Seung 2001, the 'mu' solver of scikit-learn):

    W <- W * (X H) / (W H^T H)        H <- H * (X^T W) / (H W^T W)        loss = ||X - W H^T||_F^2

restated here in float64 from an explicit start, with the same denominator floor.  Parity status: the ALGORITHM
is pinned (same start -> same factors, tests/test_models_gpu.py::test_device_nmf_matches_oracle); parity with
scikit-learn's own iterates is not claimed -- the seeded-parity path (models/hostinit.py) calls scikit-learn itself.
"""
import numpy as np

EPS = 1e-12


def nmf_mu(X, W0, H0, n_iter):
    """`n_iter` multiplicative updates from (W0, H0); returns W, H and the loss after every sweep."""
    X = np.asarray(X, dtype=np.float64)
    W = np.array(W0, dtype=np.float64)
    H = np.array(H0, dtype=np.float64)
    losses = []
    for _ in range(int(n_iter)):
        W *= (X @ H) / np.maximum(W @ (H.T @ H), EPS)
        H *= (X.T @ W) / np.maximum(H @ (W.T @ W), EPS)
        losses.append(float(((X - W @ H.T) ** 2).sum()))
    return W, H, losses

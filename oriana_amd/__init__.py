# -*- coding: utf-8 -*-
"""oriana_amd -- MI355X-native CAVI engine behind the model API of AntoinePassemiers/Oriana.

The package mirrors the reference's surface for the hot path only: ``Parameter``, ``Dimensions``,
the four factor models (``GaP``, ``ZIGaP``, ``SparseGaP``, ``SparseZIGaP``) and ``oriana.utils``.
All arithmetic runs in hand-written gfx950 kernels behind the C ABI of include/oriana_hip.h.
"""
from .exceptions import *   # noqa: F401,F403
from .dims import *         # noqa: F401,F403
from .parameters import *   # noqa: F401,F403

__version__ = '0.1'

#!/usr/bin/env python3
"""The reference's FFI on the product library (neural_net_kD of libsdpcut_nns.so, csrc/nns_compat.cpp) against the values the
real NNs.so returned for the same inputs (tests/golden/nn_k*.npz): max / median relative difference per network."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
_capi.load_library()
lib = ctypes.cdll.LoadLibrary(_capi.NNS_LIB_PATH)
for d in (2, 3, 4, 5):
    f = getattr(lib, "neural_net_%dD" % d)
    f.restype = ctypes.c_double
    g = np.load(os.path.join(G, "nn_k%d.npz" % d))
    buf = (ctypes.c_double * (d * (d + 3) // 2))()
    y = np.empty(g["nn_out"].shape[0])
    for i in range(y.shape[0]):
        buf[:] = g["inputs"][i]
        y[i] = f(buf)
    rel = np.abs(y - g["nn_out"]) / np.maximum(1.0, np.abs(g["nn_out"]))
    print("neural_net_%dD: %d inputs, bit-identical to NNs.so on %d, max |dy| / max(1, |y|) = %.2e, median %.1e"
          % (d, y.shape[0], int((y == g["nn_out"]).sum()), rel.max(), np.median(rel)))

"""Trained MLPs (2D..5D) of the optimality estimator.

The reference bakes the weights into NNs.so (source of truth neural_nets/neural_net_kD.m,
constants section).  Here they live in the data fixture ``data/nn_weights.npz`` (extracted
by tools/extract_weights.py; numbers only, provenance recorded there) and are handed to
the GPU library through ``sdpcut_set_network``.
"""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "nn_weights.npz")


def load_network(k, path=None):
    """-> (widths int32[n_layers], params float64[...]) in the packing sdpcut_set_network expects:
    xoffset, gain, ymin, (W row-major, b) per layer, y_ymin, y_gain, y_xoffset."""
    z = np.load(path or DATA)
    p = "k%d_" % k
    if p + "W1" not in z:
        raise ValueError("no trained network for %d-variable candidates" % k)
    parts = [z[p + "xoffset"].ravel(), z[p + "gain"].ravel(), np.array([z[p + "ymin"]], dtype=np.float64)]
    widths = []
    layer = 1
    while p + "W%d" % layer in z:
        W = z[p + "W%d" % layer]
        widths.append(W.shape[0])
        parts += [W.ravel(), z[p + "b%d" % layer].ravel()]
        layer += 1
    parts.append(np.array([z[p + "y_ymin"], z[p + "y_gain"], z[p + "y_xoffset"]], dtype=np.float64))
    return np.array(widths, dtype=np.int32), np.concatenate(parts).astype(np.float64)

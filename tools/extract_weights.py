#!/usr/bin/env python3
"""Extract the four trained MLPs from the reference's MATLAB text into a data fixture.

Runs ONLY in the build container (needs /root/reference).  Reads the numeric constants
of neural_nets/neural_net_{2,3,4,5}D.m (reference file: constants at the top of each
file, e.g. neural_net_3D.m:9-32) and writes

    sdpcutsel_via_nn_amd/data/nn_weights.npz

Keys, for k in 2..5:  k{k}_xoffset, k{k}_gain, k{k}_ymin  (input mapminmax, length d_in)
                      k{k}_W{l} (row-major [out, in]), k{k}_b{l}  for l = 1..n_layers
                      k{k}_y_ymin, k{k}_y_gain, k{k}_y_xoffset     (output mapminmax)

PROVENANCE: the numbers are trained weights published by rb2309/SDPCutSel-via-NN
(GPLv3, neural_nets/*.m).  Only numeric data is extracted, no code.
"""
import os
import re
import sys
import numpy as np

REF = os.environ.get("SDPCUT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "sdpcutsel_via_nn_amd", "data", "nn_weights.npz")


def parse_matrix(txt):
    txt = txt.strip()
    if txt.startswith("["):
        rows = txt[1:-1].split(";")
        return np.array([[float(v) for v in r.split()] for r in rows], dtype=np.float64)
    return np.array([[float(txt)]], dtype=np.float64)


def parse_net(path):
    src = open(path).read()
    head = src.split("% ===== SIMULATION")[0]
    assign = dict(re.findall(r"^([A-Za-z0-9_.]+)\s*=\s*(.+?);\s*$", head, flags=re.M))
    out = {}
    out["xoffset"] = parse_matrix(assign["x1_step1.xoffset"]).ravel()
    out["gain"] = parse_matrix(assign["x1_step1.gain"]).ravel()
    out["ymin"] = np.float64(assign["x1_step1.ymin"])
    out["y_ymin"] = np.float64(assign["y1_step1.ymin"])
    out["y_gain"] = np.float64(assign["y1_step1.gain"])
    out["y_xoffset"] = np.float64(assign["y1_step1.xoffset"])
    layer = 1
    while "b%d" % layer in assign:
        wname = "IW1_1" if layer == 1 else "LW%d_%d" % (layer, layer - 1)
        out["W%d" % layer] = parse_matrix(assign[wname])
        out["b%d" % layer] = parse_matrix(assign["b%d" % layer]).ravel()
        assert out["W%d" % layer].shape[0] == out["b%d" % layer].shape[0]
        layer += 1
    return out


def main():
    blob = {}
    for k in (2, 3, 4, 5):
        net = parse_net(os.path.join(REF, "neural_nets", "neural_net_%dD.m" % k))
        d_in = k * (k + 3) // 2
        assert net["xoffset"].shape == (d_in,) and net["W1"].shape[1] == d_in
        shapes = []
        layer = 1
        while "W%d" % layer in net:
            shapes.append(net["W%d" % layer].shape)
            layer += 1
        print("k=%d layers: %s" % (k, shapes))
        for name, val in net.items():
            blob["k%d_%s" % (k, name)] = val
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez(OUT, **blob)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())

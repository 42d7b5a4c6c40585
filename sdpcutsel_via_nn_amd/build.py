"""In-tree build of libsdpcut_hip.so (gfx950 only) with hipcc.

    python -m sdpcutsel_via_nn_amd.build [--force]

Objects and the shared library are written next to the sources (git-ignored, but they do
travel to the GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsdpcut_hip.so")
SOURCES = ["score.hip", "rank.hip", "topk.hip", "tri.hip", "shard.hip", "capi.hip", "cover.cpp"]
HEADERS = ["common.h", "jacobi.h", "keys.h", "topk_dev.h", os.path.join("..", "..", "include", "sdpcut.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-unused-value"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr, flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stdout=sys.stderr)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr, flush=True)
        subprocess.check_call(cmd, stdout=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

#!/bin/bash
# Round 4: the lambda_min solver (csrc/lmin.h, Householder + Laguerre) against cyclic Jacobi for everybody (-DSDPCUT_LMIN=0),
# same sources otherwise: kernel times of the eigenvalue-only kernel and of the scoring kernels, k = 2..5, 1e6 candidates.
# Build both variants in the build container first:  tools/gpu_lmin_ab.sh build     then on the GPU box:  tools/gpu_lmin_ab.sh
set -e
cd "$(dirname "$0")/.."
P=sdpcutsel_via_nn_amd
if [ "$1" = "build" ]; then
  mkdir -p $P/_abl
  python -m $P.build >/dev/null 2>&1
  for src in score eig; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DSDPCUT_LMIN=0 -c $P/csrc/$src.hip -o $P/_abl/${src}_jacobi.o &
  done
  wait
  objs=$(ls $P/csrc/*.o | grep -v "/score.o" | grep -v "/eig.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/_abl/lib_jacobi.so $P/_abl/score_jacobi.o $P/_abl/eig_jacobi.o $objs
  echo built $P/_abl/lib_jacobi.so
  exit 0
fi
for name in default jacobi; do
  lib=$P/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=$P/libsdpcut_hip.so
  echo "== $name"
  for k in 2 3 4 5; do
    SDPCUT_LIB=$PWD/$lib python3 tools/ablate.py $k 1000000 100 mfma eig,nn,eig+nn 2>/dev/null
    SDPCUT_LIB=$PWD/$lib python3 tools/eig_ab.py $k 1000000 2>/dev/null | head -1
  done
done

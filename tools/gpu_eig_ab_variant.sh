#!/bin/bash
# eigenvalue-only kernel and scoring kernel, this build against a variant library, alternating on one box: tools/gpu_eig_ab_variant.sh <variant>
v=$1
for rep in 1 2; do
for name in default $v; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  echo "== $name (pass $rep)"
  for k in 3 4 5; do
    SDPCUT_LIB=$PWD/$lib python3 tools/eig_ab.py $k 1000000 2>/dev/null | head -1
    SDPCUT_LIB=$PWD/$lib python3 tools/ablate.py $k 1000000 100 mfma eig+nn 2>/dev/null
  done
done
done

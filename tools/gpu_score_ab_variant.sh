#!/bin/bash
# scoring kernels (1e6 candidates, k = 2..5), this build against a variant library, alternating on one box: tools/gpu_score_ab_variant.sh <variant>
v=$1
for rep in 1 2 3; do
for name in default $v; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  echo "== $name (pass $rep)"
  for k in 2 3 4 5; do
    SDPCUT_LIB=$PWD/$lib python3 tools/ablate.py $k 1000000 100 mfma eig+nn 2>/dev/null
  done
done
done

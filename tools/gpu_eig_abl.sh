#!/bin/bash
# eigenvalue-only kernel variants (ABL_SRC=eig tools/build_ablation.sh ...): kernel times per size; usage tools/gpu_eig_abl.sh name1 name2 ...
for name in "$@"; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  echo "== $name"
  for k in 2 3 4 5; do SDPCUT_LIB=$PWD/$lib python3 tools/ablate.py $k 1000000 100 mfma eig 2>/dev/null; done
done

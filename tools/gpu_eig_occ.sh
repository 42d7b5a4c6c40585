#!/bin/bash
# eigenvalue-only kernel at other occupancies (ABL_SRC=eig tools/build_ablation.sh w3_6:EIG_W3=6 ...): kernel time per size
for name in default "$@"; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  echo "== $name"
  for k in 2 3 4 5; do SDPCUT_LIB=$PWD/$lib python3 tools/eig_ab.py $k 1000000 2>/dev/null | head -1; done
done

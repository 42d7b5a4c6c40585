#!/bin/bash
# A/B of library variants (sdpcutsel_via_nn_amd/_abl/lib_<name>.so) at a generic point (bench list: eigenvalue kernel + scoring
# kernels, k = 3, 4, 5) and at a structured one (bench.py --config c3: spar125-075-1 dim 4, LP points of rounds 2 and 8)
for name in default "$@"; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  echo "== $name"
  for k in 3 4 5; do
    SDPCUT_LIB=$PWD/$lib python3 tools/eig_ab.py $k 1000000 2>/dev/null | head -1
    SDPCUT_LIB=$PWD/$lib python3 tools/ablate.py $k 1000000 100 mfma eig+nn 2>/dev/null
  done
  SDPCUT_LIB=$PWD/$lib python3 bench.py --config c3 --steps 40 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin)['c3']; print('c3 strategy 4:', {k: round(v,4) for k,v in d['strategy_4'].items() if k.endswith('_ms')}); print('c3 strategy 1:', {k: round(v,4) for k,v in d['strategy_1'].items() if k.endswith('_ms')})"
done

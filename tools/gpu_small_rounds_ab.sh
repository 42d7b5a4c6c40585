#!/bin/bash
# short-list rounds, this build against a variant library (alternating, same box): tools/gpu_small_rounds_ab.sh <variant name>
v=$1
for rep in 1 2; do
for name in default $v; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$name.so
  [ "$name" = "default" ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  echo "== $name (pass $rep)"
  for f in q_50_25_75_1 q_40_8_25_1 q_30_6_50_1 q_20_20_100_2; do
    SDPCUT_LIB=$PWD/$lib python3 tools/qcqp3_time.py tests/golden/qcqp_rounds_${f}_s4.npz 2 300 2>/dev/null | head -1
  done
  SDPCUT_LIB=$PWD/$lib python3 tools/round_time.py tests/golden/rounds_spar070_050_1_d5_s4.npz 2 2>/dev/null | tail -1
  SDPCUT_LIB=$PWD/$lib python3 tools/round_time.py tests/golden/rounds_spar070_050_1_d5_s4.npz 9 2>/dev/null | tail -1
done
done

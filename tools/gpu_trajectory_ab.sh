#!/bin/bash
# separation time per recorded round, this build against -DSDPCUT_LMIN=0 (cyclic Jacobi for everybody: rounds 1-3), tools/gpu_lmin_ab.sh build first
for f in rounds_spar125_075_1_d4_s4 rounds_spar125_075_1_d3_s2 rounds_spar100_050_1_d5_s4 rounds_spar070_050_1_d5_s4 rounds_spar080_075_1_d4_s1; do
  echo "lmin   : $(python3 tools/trajectory_times.py $f 2>/dev/null)"
  echo "jacobi : $(SDPCUT_LIB=$PWD/sdpcutsel_via_nn_amd/_abl/lib_jacobi.so python3 tools/trajectory_times.py $f 2>/dev/null)"
done

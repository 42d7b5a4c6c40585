#!/bin/bash
# separation time per recorded round: this build (absolute deflation at the stopping tolerance, csrc/lmin.h) against the relative rule
# alone (-DSDPCUT_LMIN_ABS_SPLIT=0.0 -> _abl/lib_relsplit.so) and against cyclic Jacobi (_abl/lib_jacobi.so), same box
for f in rounds_spar125_075_1_d4_s4 rounds_spar125_075_2_d3_s4 rounds_spar100_050_1_d5_s4 rounds_spar080_075_1_d4_s1; do
  echo "abs+rel : $(python3 tools/trajectory_times.py $f 2>/dev/null)"
  echo "rel only: $(SDPCUT_LIB=$PWD/sdpcutsel_via_nn_amd/_abl/lib_relsplit.so python3 tools/trajectory_times.py $f 2>/dev/null)"
  echo "jacobi  : $(SDPCUT_LIB=$PWD/sdpcutsel_via_nn_amd/_abl/lib_jacobi.so python3 tools/trajectory_times.py $f 2>/dev/null)"
done

#!/bin/bash
# round_csr_kernel variants (tools/build_ablation.sh with ABL_SRC=rows): kernel averages from a kernel trace of tools/round_time.py
# usage: tools/csr_abl.sh name1 name2 ...   (default = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  lib=sdpcutsel_via_nn_amd/_abl/lib_$v.so; [ $v = default ] && lib=sdpcutsel_via_nn_amd/libsdpcut_hip.so
  [ -e $lib ] || continue
  for spec in "rounds_spar070_050_1_d5_s4 9" "rounds_spar100_050_1_d5_s4 2" "rounds_spar125_075_1_d4_s4 8"; do
    set -- $spec
    rm -rf gpurun_out/prof_csrabl
    SDPCUT_LIB=$PWD/$lib timeout -k 5 90 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_csrabl -o prof -f csv -- python3 tools/round_time.py tests/golden/$1.npz $2 60 > /dev/null 2>&1
    f=$(find gpurun_out/prof_csrabl -name prof_kernel_stats.csv | head -1)
    echo "$v $1 round $2: $(grep round_csr_kernel $f | awk -F, '{printf "round_csr_kernel avg %.1f us", $4/1000}')"
  done
done

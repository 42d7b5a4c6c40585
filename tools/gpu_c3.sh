#!/bin/bash
# Profiles of the c3 rounds (spar125-075-1 dim 4) on the GPU box -> gpurun_out/<tag>_c3_*
# usage: tools/gpu_c3.sh <tag>
set -o pipefail
export TMPDIR=/tmp
tag=$1
out=gpurun_out
mkdir -p $out
python3 bench.py --config c3 --steps 100 > $out/${tag}_c3_bench.json 2> $out/${tag}_c3_bench.err || { tail -20 $out/${tag}_c3_bench.err; exit 1; }
for route in dropin_pair fused_csr fused_rows; do
  for strat in 4 1; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_${route}_$strat -o prof -- python3 tools/c3_rounds.py $route $strat 60 > $out/${tag}_c3_${route}_s$strat.txt 2> $out/prof_${tag}_${route}_$strat.err || { tail -20 $out/prof_${tag}_${route}_$strat.err; exit 1; }
    python3 tools/timeline_rounds.py $out/prof_${tag}_${route}_$strat 1 >> $out/${tag}_c3_${route}_s$strat.txt
  done
done
for strat in 4 1; do
  python3 tools/c3_rounds.py dropin_pair $strat 200 --cprofile > $out/${tag}_c3_host_profile_s$strat.txt 2>&1
  python3 tools/c3_rounds.py dropin_pair $strat 100 --legacy --cprofile > $out/${tag}_c3_host_profile_legacy_s$strat.txt 2>&1
done
cat $out/${tag}_c3_bench.json

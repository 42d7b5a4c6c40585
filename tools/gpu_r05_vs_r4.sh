set -o pipefail
export TMPDIR=/tmp
# HEAD against the round-4 tree (git archive 46f8237 in _r4/, built in the container) on ONE box, alternating:
#   the c3 rounds and the default bench without its CPU legs
for i in 1 2; do
  (cd _r4 && timeout -k 10 200 python bench.py --config c3 --steps 60) > gpurun_out/r05_vs_r4_c3_r4_$i.json 2> gpurun_out/r05_vs_r4.err || exit 1
  timeout -k 10 200 python bench.py --config c3 --steps 60 > gpurun_out/r05_vs_r4_c3_head_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
  timeout -k 10 200 python bench.py --config c3 --steps 60 --no-prefilter > gpurun_out/r05_vs_r4_c3_headoff_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
  (cd _r4 && timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cpu-baseline) > gpurun_out/r05_vs_r4_c2_r4_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
  timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cold --no-cpu-baseline > gpurun_out/r05_vs_r4_c2_head_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
done
echo done

set -o pipefail
export TMPDIR=/tmp
# HEAD against the round-4 tree on ONE box, alternating.  _r4/ is not kept in the repository: mkdir _r4 && git archive 46f8237 | tar -x -C _r4 &&
#   (cd _r4 && python -m sdpcutsel_via_nn_amd.build) && mkdir -p _r4/tools && cp tools/trajectory_times.py _r4/tools/   (built in the container; it travels with the gpurun snapshot)
#   the c3 rounds and the default bench without its CPU legs
for i in 1 2; do
  (cd _r4 && timeout -k 10 200 python bench.py --config c3 --steps 60) > gpurun_out/r05_vs_r4_c3_r4_$i.json 2> gpurun_out/r05_vs_r4.err || exit 1
  timeout -k 10 200 python bench.py --config c3 --steps 60 > gpurun_out/r05_vs_r4_c3_head_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
  timeout -k 10 200 python bench.py --config c3 --steps 60 --no-prefilter > gpurun_out/r05_vs_r4_c3_headoff_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
  (cd _r4 && timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cpu-baseline) > gpurun_out/r05_vs_r4_c2_r4_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
  timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cold --no-cpu-baseline > gpurun_out/r05_vs_r4_c2_head_$i.json 2>> gpurun_out/r05_vs_r4.err || exit 1
done
echo done

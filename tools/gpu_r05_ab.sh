set -o pipefail
export TMPDIR=/tmp
# A/B of the fine histogram (SDPCUT_OPT_PREFILTER) on one box: the default bench twice each way, the c3 rounds each way, a kernel timeline
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cold --no-cpu-baseline > gpurun_out/r05_bench_d_on$i.json 2> gpurun_out/r05_bench_d.err || exit 1
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cold --no-cpu-baseline --no-prefilter > gpurun_out/r05_bench_d_off$i.json 2>> gpurun_out/r05_bench_d.err || exit 1
done
timeout -k 10 200 python bench.py --config c3 --steps 60 > gpurun_out/r05_c3_on.json 2>> gpurun_out/r05_bench_d.err || exit 1
timeout -k 10 200 python bench.py --config c3 --steps 60 --no-prefilter > gpurun_out/r05_c3_off.json 2>> gpurun_out/r05_bench_d.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05d -o prof -- python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 5 > gpurun_out/r05_prof_d.json 2> gpurun_out/prof_r05d.err || { tail -5 gpurun_out/prof_r05d.err; exit 1; }
python3 tools/timeline.py gpurun_out/prof_r05d > gpurun_out/r05_d_step_timeline.txt; tail -14 gpurun_out/r05_d_step_timeline.txt

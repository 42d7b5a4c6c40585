set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -s -k "direct" > gpurun_out/r05_gputests_d.log 2>&1; rc=$?; grep -v "^$" gpurun_out/r05_gputests_d.log | tail -20; echo pytest rc=$rc
if [ $rc -ge 124 ]; then exit $rc; fi
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cpu-baseline > gpurun_out/r05_bench_d_on$i.json 2> gpurun_out/r05_bench_d.err || exit 1
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cpu-baseline --no-prefilter > gpurun_out/r05_bench_d_off$i.json 2>> gpurun_out/r05_bench_d.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05d -o prof -- python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 5 > gpurun_out/r05_prof_d.json 2> gpurun_out/prof_r05d.err || { tail -5 gpurun_out/prof_r05d.err; exit 1; }
python3 tools/timeline.py gpurun_out/prof_r05d > gpurun_out/r05_d_step_timeline.txt; tail -14 gpurun_out/r05_d_step_timeline.txt

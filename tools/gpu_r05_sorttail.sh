set -o pipefail
export TMPDIR=/tmp
# the selection tail (refine, tile sort, merge ranks): parity, then bench + kernel timeline on the same box
timeout -k 10 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -x -q > gpurun_out/r05_sorttail_tests.txt 2>&1 || { tail -30 gpurun_out/r05_sorttail_tests.txt; exit 1; }
tail -2 gpurun_out/r05_sorttail_tests.txt
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-c3 --no-cold --no-cpu-baseline > gpurun_out/r05_bench_st$i.json 2> gpurun_out/r05_bench_st.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05st -o prof -- python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 5 > gpurun_out/r05_prof_st.json 2> gpurun_out/prof_r05st.err || { tail -5 gpurun_out/prof_r05st.err; exit 1; }
python3 tools/timeline.py gpurun_out/prof_r05st > gpurun_out/r05_st_step_timeline.txt; tail -16 gpurun_out/r05_st_step_timeline.txt

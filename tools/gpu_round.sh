#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-r01}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/test_$tag.log 2>&1
rc=$?
tail -5 gpurun_out/test_$tag.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -20 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof_$tag.json 2> gpurun_out/prof_$tag.err || { tail -20 gpurun_out/prof_$tag.err; exit 1; }
find gpurun_out/prof_$tag -name "*stats*" | head

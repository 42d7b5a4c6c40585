#!/bin/bash
# Kernel + memory-copy timeline of bench steps (rocprofv3, no counters).  usage: tools/gpu_timeline.sh <tag> [bench args...]
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof_$tag -o prof -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary "$@" > gpurun_out/bench_prof_$tag.json 2> gpurun_out/prof_$tag.err || { tail -20 gpurun_out/prof_$tag.err; exit 1; }
python3 tools/timeline.py gpurun_out/prof_$tag > gpurun_out/timeline_$tag.txt
cat gpurun_out/timeline_$tag.txt

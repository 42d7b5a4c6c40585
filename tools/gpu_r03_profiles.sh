#!/bin/bash
# Round-3 evidence in one GPU-box session -> gpurun_out/r03_* (copy what is to be judged into profiles/).
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
step() { echo "=== $*" >&2; }
step "bench c2 (driver form) + kernel stats + PMC + traffic"
python3 bench.py > $out/r03_bench.json 2> $out/r03_bench.err || { tail -5 $out/r03_bench.err; exit 1; }
tools/gpu_profile.sh r03_k3 > $out/r03_k3_profile.log 2>&1 || { tail -5 $out/r03_k3_profile.log; exit 1; }
step "c3 rounds"
tools/gpu_c3.sh r03 > $out/r03_c3.log 2>&1 || { tail -5 $out/r03_c3.log; exit 1; }
step "eigenvalue-only kernel: sizes, PMC"
for k in 2 3 4 5; do python3 tools/ablate.py $k 1000000 100 mfma eig; done > $out/r03_eig_kernel_all_sizes.txt 2>/dev/null
python3 tools/eig_ab.py 3 1000000 >> $out/r03_eig_kernel_all_sizes.txt 2>/dev/null
python3 tools/eig_ab.py 4 1000000 >> $out/r03_eig_kernel_all_sizes.txt 2>/dev/null
tools/gpu_pmc.sh r03_eigpmc_k3 3 1000000 mfma eig eig_only > $out/r03_eig_k3_pmc.log 2>&1
tools/gpu_pmc.sh r03_eigpmc_k4 4 1000000 mfma eig eig_only > $out/r03_eig_k4_pmc.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/r03_eigtraffic_$c -o pmc -- python3 tools/ablate.py 3 1000000 100 mfma eig > $out/r03_eigtraffic_$c.log 2>&1
done
python3 tools/pmc_summary.py eig_only $out/r03_eigtraffic_FETCH_SIZE $out/r03_eigtraffic_WRITE_SIZE > $out/r03_eig_k3_hbm_traffic.txt
step "config 4 shard"
python3 bench.py --config c4-shard --no-cpu-baseline --steps 50 > $out/r03_bench_c4_shard.json 2>/dev/null
python3 bench.py --config c4-shard --no-cpu-baseline --steps 50 --no-pinned-point > $out/r03_bench_c4_shard_pageable_point.json 2>/dev/null
tools/gpu_timeline.sh r03_c4shard --config c4-shard > /dev/null 2>&1; cp $out/timeline_r03_c4shard.txt $out/r03_c4shard_step_timeline.txt
step "sharded code path at N = 1"
SDPCUT_BENCH_FORCE_SHARDED=1 python3 bench.py --no-cpu-baseline --no-secondary > $out/r03_bench_forced_sharded.json 2>/dev/null
SDPCUT_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --no-cpu-baseline --no-secondary > $out/r03_bench_one_rank_rccl.json 2>/dev/null
step "accuracy"
python3 tools/accuracy.py > $out/r03_accuracy.txt 2>/dev/null
python3 tools/compat_accuracy.py >> $out/r03_accuracy.txt 2>/dev/null
step "variants"
if [ -e sdpcutsel_via_nn_amd/_abl/lib_j1.so ]; then
  tools/gpu_abl_bench.sh default j1 > $out/r03_mfma_three_waves.txt 2>/dev/null
  SDPCUT_LIB=$PWD/sdpcutsel_via_nn_amd/_abl/lib_j1.so tools/gpu_pmc.sh r03_j1pmc 3 1000000 mfma eig+nn score_mfma >> $out/r03_mfma_three_waves.txt 2>&1
  tools/gpu_pmc.sh r03_j2pmc 3 1000000 mfma eig+nn score_mfma > $out/r03_mfma_two_waves_pmc.txt 2>&1
fi
if [ -e sdpcutsel_via_nn_amd/_abl/lib_eignopack.so ]; then
  tools/gpu_eig_abl.sh default eignopack > $out/r03_eig_kernel_variants.txt 2>/dev/null
fi
step done

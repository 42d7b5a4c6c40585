#!/bin/bash
# Round-5 evidence in GPU-box sessions -> gpurun_out/r05_* (copy what is to be judged into profiles/).
#   tools/gpu_r05_profiles.sh a    bench (driver form), kernel stats + PMC + HBM traffic of the bench command, eigenvalue kernel PMC
#   tools/gpu_r05_profiles.sh b    c3 cover: PMC of score_mfma_all_kernel at the LP points of rounds 2 and 4, k = 4 synthetic PMC, c3 bench
#   tools/gpu_r05_profiles.sh c    config-4 shard, sharded path at N = 1, one-rank RCCL, two self-launched ranks on one GPU, accuracy
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
step() { echo "=== $*" >&2; }
GROUPS3=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA" \
 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_IFETCH")
case "$1" in
a)
  step "bench c2 (driver form)"
  python3 bench.py > $out/r05_bench.json 2> $out/r05_bench.err || { tail -5 $out/r05_bench.err; exit 1; }
  step "kernel stats + PMC + traffic of the bench command"
  tools/gpu_profile.sh r05_k3 > $out/r05_k3_profile.log 2>&1 || { tail -5 $out/r05_k3_profile.log; exit 1; }
  step "eigenvalue-only kernel PMC"
  tools/gpu_pmc.sh r05_eigpmc_k3 3 1000000 mfma eig eig_only > $out/r05_eig_k3_pmc.log 2>&1
  cp $out/r05_eigpmc_k3_summary.txt $out/r05_eig_k3_kernel_pmc.txt
  ;;
b)
  step "c3: PMC of the scoring launch at the LP points of rounds 2 (next to the McCormick vertex) and 4 (generic)"
  for r in 2 4; do
    i=0
    for grp in "${GROUPS3[@]}"; do
      i=$((i+1))
      rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_r05_c3r${r}_$i -o pmc -- python3 tools/c3_rounds.py fused_csr 4 30 --round $r > $out/pmc_r05_c3r${r}_$i.log 2>&1 || { tail -5 $out/pmc_r05_c3r${r}_$i.log; exit 1; }
    done
    python3 tools/pmc_summary.py "score_mfma_kernel<4" $out/pmc_r05_c3r${r}_1 $out/pmc_r05_c3r${r}_2 $out/pmc_r05_c3r${r}_3 > $out/r05_c3_round${r}_score_kernel_pmc.txt
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_r05_c3r$r -o prof -- python3 tools/c3_rounds.py fused_csr 4 60 --round $r > $out/r05_c3_round${r}_rounds.txt 2> $out/prof_r05_c3r$r.err
    python3 tools/timeline_rounds.py $out/prof_r05_c3r$r 1 >> $out/r05_c3_round${r}_rounds.txt
  done
  step "k = 4 synthetic PMC"
  tools/gpu_pmc.sh r05_k4pmc 4 1000000 mfma eig+nn score_mfma > $out/r05_k4_pmc.log 2>&1
  cp $out/r05_k4pmc_summary.txt $out/r05_k4_score_kernel_pmc.txt
  step "c3 bench"
  python3 bench.py --config c3 --steps 100 > $out/r05_c3_bench.json 2> $out/r05_c3_bench.err
  ;;
c)
  step "config 4 shard"
  python3 bench.py --config c4-shard --no-cpu-baseline --steps 50 > $out/r05_bench_c4_shard.json 2>/dev/null
  step "sharded code path at N = 1, one-rank RCCL, self-launched two ranks on one GPU"
  SDPCUT_BENCH_FORCE_SHARDED=1 python3 bench.py --no-cpu-baseline --no-secondary > $out/r05_bench_forced_sharded.json 2>/dev/null
  SDPCUT_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --no-cpu-baseline --no-secondary > $out/r05_bench_one_rank_rccl.json 2>/dev/null
  SDPCUT_BENCH_ONE_DEVICE=1 SDPCUT_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 50 --no-cpu-baseline --no-secondary > $out/r05_bench_plain_gpus2_one_device.json 2> $out/r05_bench_plain_gpus2.err
  step "accuracy"
  python3 tools/accuracy.py > $out/r05_accuracy.txt 2>/dev/null
  python3 tools/compat_accuracy.py >> $out/r05_accuracy.txt 2>/dev/null
  ;;
esac
step done

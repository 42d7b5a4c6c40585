#!/bin/bash
# Round-4 evidence in one GPU-box session -> gpurun_out/r04_* (copy what is to be judged into profiles/).
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
step() { echo "=== $*" >&2; }
step "bench c2 (driver form)"
python3 bench.py > $out/r04_bench.json 2> $out/r04_bench.err || { tail -5 $out/r04_bench.err; exit 1; }
step "kernel stats + PMC + traffic of the bench command"
tools/gpu_profile.sh r04_k3 > $out/r04_k3_profile.log 2>&1 || { tail -5 $out/r04_k3_profile.log; exit 1; }
step "k = 4 score kernel PMC"
tools/gpu_pmc.sh r04_k4pmc 4 1000000 mfma eig+nn score_mfma > $out/r04_k4_pmc.log 2>&1
cp $out/r04_k4pmc_summary.txt $out/r04_k4_score_kernel_pmc.txt
step "eigenvalue-only kernel: sizes, PMC, traffic"
for k in 2 3 4 5; do python3 tools/eig_ab.py $k 1000000 2>/dev/null | head -1; done > $out/r04_eig_kernel_all_sizes.txt
for k in 3 4 5; do
  tools/gpu_pmc.sh r04_eigpmc_k$k $k 1000000 mfma eig eig_only > $out/r04_eig_k${k}_pmc.log 2>&1
  cp $out/r04_eigpmc_k${k}_summary.txt $out/r04_eig_k${k}_kernel_pmc.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/r04_eigtraffic_$c -o pmc -- python3 tools/ablate.py 3 1000000 100 mfma eig > $out/r04_eigtraffic_$c.log 2>&1
done
python3 tools/pmc_summary.py eig_only $out/r04_eigtraffic_FETCH_SIZE $out/r04_eigtraffic_WRITE_SIZE > $out/r04_eig_k3_hbm_traffic.txt
step "lmin against Jacobi: kernels, trajectories"
tools/gpu_lmin_ab.sh > $out/r04_lmin_vs_jacobi_kernels.txt 2>&1
tools/gpu_trajectory_ab.sh > $out/r04_separation_time_per_recorded_round.txt 2>&1
step "c3 rounds"
python3 bench.py --config c3 --steps 100 > $out/r04_c3_bench.json 2> $out/r04_c3_bench.err
step "config 4 shard"
python3 bench.py --config c4-shard --no-cpu-baseline --steps 50 > $out/r04_bench_c4_shard.json 2>/dev/null
step "sharded code path at N = 1, one-rank RCCL, self-launched two ranks on one GPU"
SDPCUT_BENCH_FORCE_SHARDED=1 python3 bench.py --no-cpu-baseline --no-secondary > $out/r04_bench_forced_sharded.json 2>/dev/null
SDPCUT_FORCE_COLLECTIVES=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --no-cpu-baseline --no-secondary > $out/r04_bench_one_rank_rccl.json 2>/dev/null
SDPCUT_BENCH_ONE_DEVICE=1 SDPCUT_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 50 --no-cpu-baseline --no-secondary > $out/r04_bench_plain_gpus2_one_device.json 2> $out/r04_bench_plain_gpus2.err
step "accuracy"
python3 tools/accuracy.py > $out/r04_accuracy.txt 2>/dev/null
python3 tools/compat_accuracy.py >> $out/r04_accuracy.txt 2>/dev/null
step "noise floor of the reference's eigenvalues: this box's LAPACK against the recorded order"
for a in "rounds_spar125_075_1_d4_s4 5" "rounds_spar125_075_2_d3_s4 17 18 19"; do
  echo "--- this build"; python3 tools/lmin_diag.py $a 2>/dev/null
  echo "--- -DSDPCUT_LMIN=0 (cyclic Jacobi, rounds 1-3)"; SDPCUT_LIB=$PWD/sdpcutsel_via_nn_amd/_abl/lib_jacobi.so python3 tools/lmin_diag.py $a 2>/dev/null
done > $out/r04_lambda_min_noise_floor_gpu.txt
step done

#!/usr/bin/env python3
"""Benchmark of the cut-scoring hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one selection round over one batch of synthetic candidates, inputs resident in
HBM: copy the LP point device->device, score every candidate (Jacobi lambda_min + MLP
optimality measure), rank with the combined strategy (sel_size = 5000), merge the per-shard
heads (N > 1: RCCL all-gather), and generate the eigen-cut rows of the selected candidates.
Workload at every N: BASELINE.json configs[1] per GPU (n = 100 dense, 1e6 random 3-variable
index sets, seed 7 + rank) -> weak scaling; `value` = candidates scored per second over all
ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 10 ** 6
NB_VARS = 100
K = 3
SEL = 5000
BASELINE_METRIC = "candidate cuts scored/sec (eig+NN), 1e6 3-var subs, 1/2/4/8 GPU"    # BASELINE.json "metric"
PREWARM_STEPS = 150    # untimed setup before the W warmup steps (GPU clock ramp), see main()
FLOPS_PER_CAND = {2: 17152, 3: 11000, 4: 11500, 5: 27264}     # MLP mul+add only, BASELINE.md section 4
BYTES_PER_CAND = {2: 24, 3: 28, 4: 32, 5: 36}                 # index set in, two fp64 scores out
FP64_PEAK_TFLOPS = 78.6                                       # MI355X fp64 matrix = vector peak
HBM_PEAK_GBS = 8000.0


def cpu_baseline(wl, sample):
    """The oracle ("port": C restatement of NNs.so + batched LAPACK eigvalsh + numpy ranking),
    1 core, on the first `sample` candidates of the same workload."""
    from oracle import cutsel_oracle as oracle
    si = np.ascontiguousarray(wl["set_inds"][:sample, :K])
    vv, Q = wl["vars_values"], wl["Q_arr"]
    L = NB_VARS * (NB_VARS + 1) // 2
    t0 = time.perf_counter()
    obj = oracle.opt_score_batch(K, si, NB_VARS, vv, Q)
    lam = oracle.eigmin_batch(K, vv[L:][si], vv[:L][oracle.triu_positions(si, NB_VARS)])
    order, _, _, _ = oracle.rank_arrays(4, obj, lam, min(SEL, sample))
    for c in order[:min(SEL, sample)][:SEL]:
        w, v = oracle.get_eigendecomp(K, vv[L:][si[c]], vv[:L][oracle.triu_positions(si[c], NB_VARS)], True)
    dt = time.perf_counter() - t0
    # the reference's own shape -- a Python loop with one ctypes call
    # into the NNs.so-compatible entry point and one LAPACK eigvalsh per candidate -- on a small
    # sub-sample (optimality list + feasibility list = both measures for every candidate)
    m = min(20000, sample)
    agg = oracle.build_agg_list([tuple(int(v) for v in s) for s in si[:m]], NB_VARS, Q)
    t1 = time.perf_counter()
    oracle.sel_eigcut_by_ordering_on_measure(agg, L, 2, vv)
    oracle.sel_eigcut_by_ordering_on_measure(agg, L, 1, vv)
    dt_loop = time.perf_counter() - t1
    return dict(value=sample / dt, unit="candidates/s", cores=1, kind="port",
                sample="first %d of the %d candidates of this workload, same round (score eig+NN, combined "
                       "ranking, %d eigh cut rows), %.1f s" % (sample, N_PER_GPU, min(SEL, sample), dt),
                reference_style_loop=dict(value=m / dt_loop, unit="candidates/s", cores=1,
                                          sample="per-candidate Python loop (ctypes NN + eigvalsh each) on the "
                                                 "first %d candidates, %.1f s" % (m, dt_loop)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kernel", choices=["mfma", "simple", "valu"], default="mfma")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse", action="store_true", help="A/B: run the selection's key pass inside the score kernel")
    ap.add_argument("--cpu-sample", type=int, default=10 ** 6)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                     % (args.gpus, args.gpus))
    # rehearsal knobs (one-GPU box): SDPCUT_BENCH_BACKEND=gloo stages the all-gather through the
    # host, SDPCUT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0.  The driver's runs use neither.
    backend = os.environ.get("SDPCUT_BENCH_BACKEND", "nccl")
    # SDPCUT_BENCH_FORCE_SHARDED=1: run the N > 1 code path (packed head, finish) at N = 1 to see
    # what it costs over the fused single-GPU round, the collective itself excluded
    force_sharded = os.environ.get("SDPCUT_BENCH_FORCE_SHARDED") == "1"
    # SDPCUT_FORCE_COLLECTIVES=1 (under `torch.distributed.run --nproc-per-node 1`): rehearse the N > 1
    # path against real RCCL on a one-GPU box -- process group, barrier, all-gather, all-reduce with
    # one rank
    solo_dist = world == 1 and os.environ.get("SDPCUT_FORCE_COLLECTIVES") == "1" and "MASTER_ADDR" in os.environ
    if solo_dist:
        force_sharded = True
    if os.environ.get("SDPCUT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or solo_dist
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()

    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    from sdpcutsel_via_nn_amd.distributed import DeviceOps, ShardedSelector

    wl = synthetic.make_workload(nb_vars=NB_VARS, k=K, count=N_PER_GPU, seed=7 + rank)
    if rank != 0:
        # one LP point and one objective for the whole job (rank 0's); shards differ in index sets
        wl0 = synthetic.make_instance(NB_VARS, seed=7)
        wl["Q_arr"], wl["vars_values"] = wl0[0], wl0[1]
    sc = _capi.Scorer(local_rank)
    sc.set_option(_capi.OPT_TIMING, 1)
    if args.fuse:
        sc.set_option(_capi.OPT_FUSE_KEYS, 1)
    sc.set_option(_capi.OPT_KERNEL, {"mfma": _capi.KERNEL_MFMA, "simple": _capi.KERNEL_SIMPLE, "valu": _capi.KERNEL_VALU}[args.kernel])
    sc.set_network(K, *networks.load_network(K))
    sc.set_instance(NB_VARS, wl["Q_arr"])
    sc.set_candidates(wl["set_inds"], wl["ks"], global_base=rank * N_PER_GPU)
    d_vars = torch.from_numpy(wl["vars_values"]).to(device)
    sel = None
    if world > 1 or force_sharded:
        # DeviceOps binds the library to torch's current stream: its kernels, torch's copies and the
        # hand-off to the collective are ordered without host synchronisation (a dedicated
        # non-blocking stream measured no better at N = 1 and worse with two ranks on one GPU)
        sel = ShardedSelector(DeviceOps(sc, device), N_PER_GPU)

    kernel_ms = []

    def step():
        sc.set_point_device(d_vars.data_ptr())
        if world == 1 and not force_sharded:
            # one C-ABI call: score (eig + NN) -> combined ranking -> cut rows of the head -> one D2H
            # (results land in the handle's pinned host block; copy=False hands out views of it)
            res = rows = sc.select_round(4, SEL, copy=False)
        else:
            # score the shard -> packed head record -> ONE all-gather (RCCL) -> replicated merge ->
            # each rank generates the rows of its own candidates -> one D2H, one host sync
            sc.score(_capi.EIG | _capi.NN)
            res = rows = sel.select_round(4, SEL)
        kernel_ms.append(sc.last_timing()[0])
        return res, rows

    # Setup, untimed: the GPU comes out of idle with low clocks and needs ~50 ms of load to reach the
    # sustained state (score kernel 0.42 -> 0.39 ms); bring it there before the W warmup steps so that
    # short runs (K = 20) measure the same machine state as long ones.
    # A full (generation-2) Python garbage collection walks the millions of objects `import torch`
    # creates and stalls the host for ~75 ms once every few hundred steps: park them in the permanent
    # generation, as latency-sensitive Python services do.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(PREWARM_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    del kernel_ms[:]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, rows = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total = N_PER_GPU * world * args.steps
        k_ms = float(np.mean(kernel_ms))
        tflops = FLOPS_PER_CAND[K] * N_PER_GPU / (k_ms * 1e-3) / 1e12
        gbs = BYTES_PER_CAND[K] * N_PER_GPU / (k_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "score_kernel_traffic.json")
        if os.path.exists(tfile):      # PMC-measured HBM bytes per launch (separate rocprofv3 --pmc passes)
            traffic = json.load(open(tfile)).get(args.kernel)
        out = {
            "metric": BASELINE_METRIC, "value": total / dt, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic n=100 dense X, 1e6 random 3-var index sets per GPU, "
                                   "eig + neural_net_3D scoring, combined ranking sel_size=5000, cut rows",
                       "candidates_per_gpu": N_PER_GPU, "nb_vars": NB_VARS, "k": K, "sel_size": SEL,
                       "kernel": args.kernel, "strategy": 4},
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": {"mfma": "score_mfma_kernel<3, 50, 3, %s>" % ("true" if args.fuse else "false"), "valu": "score_valu_kernel<3,50,3>", "simple": "score_simple_kernel<3>"}[args.kernel],
                         "kernel_ms": k_ms, "flops_per_candidate": FLOPS_PER_CAND[K],
                         "hbm_algorithmic_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                         "bytes_per_candidate": BYTES_PER_CAND[K]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, min(args.cpu_sample, N_PER_GPU))
        print(json.dumps(out), flush=True)
    sc.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

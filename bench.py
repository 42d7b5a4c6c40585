#!/usr/bin/env python3
"""Benchmark of the cut-scoring hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c2|c4|c4-shard]

One "step" = one selection round over one batch of synthetic candidates whose index sets are
resident in HBM, bracketed HOST TO HOST (SURVEY.md section 8 d): the LP point starts in a host
array (`sdpcut_set_point`), every candidate is scored (lambda_min: Householder + Laguerre, csrc/lmin.h; MLP optimality
measure), ranked with the combined strategy (sel_size = 5000), the per-shard heads are merged
(N > 1: one RCCL all-gather), the eigen-cut rows of the selected candidates are generated, and
the round's results are back in host memory when the step ends.

  c2 (default)  BASELINE.json configs[1] per GPU: n = 100 dense, 1e6 random 3-variable index sets
                (seed 7 + rank), weak scaling -- the configuration the metric is quoted on;
                at N = 1 the line also carries the k = 2, 4, 5 rates as secondary fields.
  c4            BASELINE.json configs[3]: n = 1000, 1e8 3-variable candidates IN TOTAL, generated on
                the device by the counter-based Philox generator (candidate id -> index set), rank r
                takes ids [r, r+1) * 1e8 / N: strong scaling.
  c4-shard      one rank's share of the 8-GPU form of c4 (1.25e7 candidates per GPU).
  c3            BASELINE.json configs[2] as real callers run it: the dim-4 cover of spar125-075-1 (1 700 215 mixed
                2/3/4-variable sets, enumerated on the device) at two LP points the reference's own trajectory recorded
                (tests/golden/rounds_spar125_075_1_d4_s4.npz: round 2 = combined strategy, round 8 = feasibility), through
                the fused C-ABI calls AND through the drop-in pair _sel_eigcut_by_ordering_on_measure + _gen_eigcuts_selected
                (cut_select_qp.py:165-182).  The default (c2) run carries the same numbers as secondary.c3.

`value` = candidates scored per second over all ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEL = 5000
BASELINE_METRIC = "candidate cuts scored/sec (eig+NN), 1e6 3-var subs, 1/2/4/8 GPU"    # BASELINE.json "metric"
PREWARM_STEPS = 150    # untimed setup before the W warmup steps (GPU clock ramp), see main()
FLOPS_PER_CAND = {2: 17152, 3: 11000, 4: 11500, 5: 27264}     # MLP mul+add only, BASELINE.md section 4
BYTES_PER_CAND = {2: 24, 3: 28, 4: 32, 5: 36}                 # index set in, two fp64 scores out
FP64_PEAK_TFLOPS = 78.6                                       # MI355X fp64 matrix = vector peak
HBM_PEAK_GBS = 8000.0
KERNEL_SHAPES = {2: "2, 64, 3", 3: "3, 50, 3", 4: "4, 50, 3", 5: "5, 64, 4"}     # K, hidden width, hidden layers


def kernel_name(k, counts_digit, variant="mfma"):
    """rocprofv3's name of the scoring kernel of size k that this run launches.  MFMA variant: <K, H, NH, FUSE, CLAMP, J>; FUSE = 3
    (TK_MODE_STRONG) when the kernel also counts the leading digit of the combined strategy's selection keys (the fused round and the
    sharded round alike; 0 with --no-fuse-keys), CLAMP = false for the shipped networks (pre-activations provably bounded), J = 2
    column tiles per pass.  The other variants carry no selection state."""
    if variant == "valu":
        return "score_valu_kernel<%s>" % KERNEL_SHAPES[k]
    if variant == "simple":
        return "score_simple_kernel<%d>" % k
    return "score_mfma_kernel<%s, %d, false, 2>" % (KERNEL_SHAPES[k], 3 if counts_digit else 0)


CONFIGS = {
    "c2": dict(nb_vars=100, k=3, total=None, per_gpu=10 ** 6, scaling="weak",
               text="configs[1]: synthetic n=100 dense X, 1e6 random 3-var index sets per GPU, eig + neural_net_3D "
                    "scoring, combined ranking sel_size=5000, cut rows; host-to-host round"),
    "c4": dict(nb_vars=1000, k=3, total=10 ** 8, per_gpu=None, scaling="strong",
               text="configs[3]: synthetic n=1000, 1e8 3-var candidates in total from the on-device Philox generator, "
                    "sharded by candidate id, RCCL all-gather of per-shard top-k; host-to-host round"),
    "c4-shard": dict(nb_vars=1000, k=3, total=None, per_gpu=12_500_000, scaling="weak",
                     text="configs[3], one of 8 shards: synthetic n=1000, 1.25e7 3-var candidates per GPU from the "
                          "on-device Philox generator; host-to-host round"),
}


# ----------------------------------------------------------------------------- CPU baseline (oracle = checker code, timed)
def _cpu_score_chunk(args):
    """worker of the all-cores leg: both measures for one chunk of candidates (oracle code)"""
    k, nb_vars, si, vv, Q = args
    from oracle import cutsel_oracle as oracle
    L = nb_vars * (nb_vars + 1) // 2
    obj = oracle.opt_score_batch(k, si, nb_vars, vv, Q)
    lam = oracle.eigmin_batch(k, vv[L:][si], vv[:L][oracle.triu_positions(si, nb_vars)])
    return obj, lam


def host_cores():
    """-> (cores this process may use, how that was found): the scheduler affinity mask, cut by the cgroup's CPU quota when there
    is one (a container on a 256-thread host may be allowed all 256 or a 16-CPU share; oversubscribing a quota only adds
    throttling)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    cores = aff if quota is None else max(1, min(aff, int(quota)))
    return cores, "len(os.sched_getaffinity(0)) = %d, cgroup CPU quota = %s, os.cpu_count() = %d" % (
        aff, "none" if quota is None else "%.1f" % quota, os.cpu_count() or 0)


def selection_parity(oracle, strat, obj, lam, max_elem, sel, gpu):
    """The GPU round against the ranking of the ORACLE'S OWN scores on the same list (cut_select_qp.py:601-632, :639-654).
    gpu: dict(idx, score, new_strat, eig, obj) -- head ids / scores of one round as the device returned them and the device's
    score arrays (None where the round did not compute the measure).  Position-by-position comparison of the head; the score
    deviations are over EVERY candidate of the list, not the head only."""
    order, scores, new_strat, _ = oracle.rank_arrays(strat, obj, lam, sel)
    head, head_scores = order[:sel], scores[:sel]
    ids = np.asarray(gpu["idx"], dtype=np.int64)
    m = min(len(head), len(ids))
    differing = int((head[:m] != ids[:m]).sum()) + abs(len(head) - len(ids))
    one_side = int(len(np.setxor1d(head, ids)))
    out = {"strategy": strat, "head": int(len(head)), "gpu_head": int(len(ids)), "topk_identical": bool(differing == 0),
           "positions_differing": differing, "ids_one_side_only": one_side,
           "new_strategy_identical": bool(int(gpu["new_strat"]) == int(new_strat)),
           "against": "oracle.rank_arrays on the oracle's own obj_improve / lambda_min of all %d candidates" % len(order if strat != 1 else lam)}
    if m:
        same = head[:m] == ids[:m]
        out["max_abs_d_head_score"] = float(np.abs(np.asarray(gpu["score"])[:m][same] - head_scores[:m][same]).max()) if same.any() else None
    if gpu.get("eig") is not None and lam is not None:
        out["max_abs_d_eig"] = float(np.abs(gpu["eig"] - lam).max())
    if gpu.get("obj") is not None and obj is not None:
        # the tolerance form of the parity tests (hard part 5): |d| relative to max(|score|, 1e-3 max_elem)
        out["max_rel_d_obj"] = float((np.abs(gpu["obj"] - obj) / np.maximum(np.abs(obj), 1e-3 * max_elem)).max())
        out["max_abs_d_obj"] = float(np.abs(gpu["obj"] - obj).max())
    return out


def cpu_baseline(set_inds, nb_vars, k, vv, Q, n_workload, sample, gpu_rounds=None):
    """The oracle ("port": C restatement of NNs.so + batched LAPACK eigvalsh + numpy ranking) on the
    first `sample` candidates of the same workload: 1 core, all cores this process may use (process pool), and the
    reference's own shape (per-candidate Python loop).  gpu_rounds {strategy: record of selection_parity}: when the sample is
    the whole list, the GPU rounds of this run are compared with the ranking of the oracle's scores -> out["parity"]."""
    from oracle import cutsel_oracle as oracle
    si = np.ascontiguousarray(set_inds[:sample, :k])
    L = nb_vars * (nb_vars + 1) // 2
    sel = min(SEL, sample)

    def finish(obj, lam):
        order, _, _, _ = oracle.rank_arrays(4, obj, lam, sel)
        for c in order[:sel]:
            oracle.get_eigendecomp(k, vv[L:][si[c]], vv[:L][oracle.triu_positions(si[c], nb_vars)], True)

    t0 = time.perf_counter()
    obj1, lam1 = _cpu_score_chunk((k, nb_vars, si, vv, Q))
    finish(obj1, lam1)
    dt = time.perf_counter() - t0
    out = dict(value=sample / dt, unit="candidates/s", cores=1, kind="port",
               sample="first %d of the %d candidates of this workload, same round (score eig+NN, combined "
                      "ranking, %d eigh cut rows), %.1f s" % (sample, n_workload, sel, dt))
    if gpu_rounds and sample == n_workload:
        q = np.abs(np.asarray(Q)[oracle.triu_positions(si, nb_vars)]).max(axis=1) * k
        max_elem = np.where(q == 0, 1.0, q)
        par = {}
        for strat, rec in sorted(gpu_rounds.items()):
            par[strat] = selection_parity(oracle, strat, None if strat == 1 else obj1, None if strat == 2 else lam1, max_elem, sel, rec)
        main = dict(par.get(4) or par[sorted(par)[0]])
        for strat, p in par.items():
            if p is not main and strat != main["strategy"]:
                main["strategy_%d" % strat] = p
        out["parity"] = main
    # all host cores this process may use (SURVEY 8 d): the scoring split over a process pool;
    # the ranking and the cut rows (serial numpy / LAPACK) are timed separately
    cores, how = host_cores()
    if cores > 1:
        import multiprocessing as mp
        chunks = np.array_split(np.arange(sample), cores * 4)
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_cpu_score_chunk, [(k, nb_vars, si[:64], vv, Q)] * cores)          # start the workers (untimed)
            t1 = time.perf_counter()
            parts = pool.map(_cpu_score_chunk, [(k, nb_vars, si[c], vv, Q) for c in chunks])
            obj_all, lam_all = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
            t2 = time.perf_counter()
            finish(obj_all, lam_all)
            t3 = time.perf_counter()
        out["all_cores"] = dict(value=sample / (t2 - t1), unit="candidates/s", cores=cores,
                                scoring_s=t2 - t1, ranking_and_rows_s_one_core=t3 - t2, value_whole_round=sample / (t3 - t1),
                                sample="same sample, eig + NN scoring over a %d-process pool (%s): %.2f s; `value` is the scoring "
                                       "rate; the ranking and the %d cut rows run on one core after it: %.2f s"
                                       % (cores, how, t2 - t1, sel, t3 - t2))
        out["all_cores_value"], out["all_cores_cores"] = out["all_cores"]["value"], cores      # (scalars survive the driver's record)
    # the reference's own shape -- a Python loop with one ctypes call
    # into the NNs.so-compatible entry point and one LAPACK eigvalsh per candidate -- on a small
    # sub-sample (optimality list + feasibility list = both measures for every candidate)
    m = min(100000, sample)      # SURVEY 8 d: N = 1e5 subsample
    agg = oracle.build_agg_list([tuple(int(v) for v in s) for s in si[:m]], nb_vars, Q)
    t1 = time.perf_counter()
    oracle.sel_eigcut_by_ordering_on_measure(agg, L, 2, vv)
    oracle.sel_eigcut_by_ordering_on_measure(agg, L, 1, vv)
    dt_loop = time.perf_counter() - t1
    out["reference_style_loop"] = dict(value=m / dt_loop, unit="candidates/s", cores=1,
                                       sample="per-candidate Python loop (ctypes NN + eigvalsh each) on the first %d "
                                              "candidates, %.1f s" % (m, dt_loop))
    return out


# ----------------------------------------------------------------------------- c3: the round as real callers run it
C3_INSTANCE, C3_DIM, C3_ROUNDS = "spar125-075-1", 4, {4: 2, 1: 8}      # strategy -> recorded round whose LP point is replayed


def c3_setup(device_index):
    """-> (CutSolver bound to the device-enumerated cover, its Scorer, {strategy: LP point}, candidate count)"""
    from sdpcutsel_via_nn_amd import harness
    from sdpcutsel_via_nn_amd.cut_solver import CutSolver, DeviceAgg
    gold = os.path.join(ROOT, "tests", "golden")
    inst = harness.parse_boxqp(os.path.join(gold, "instances", C3_INSTANCE + ".in"))
    g = np.load(os.path.join(gold, "rounds_%s_d%d_s4.npz" % (C3_INSTANCE.replace("-", "_"), C3_DIM)))
    pts = {}
    for strat, r in C3_ROUNDS.items():
        assert int(g["r%02d_strat" % r]) == strat
        pts[strat] = np.ascontiguousarray(g["r%02d_vars" % r], dtype=np.float64)
    cs = CutSolver(device_index)
    cs._dim, cs._nb_vars, cs._nb_lifted, cs._Q_arr = C3_DIM, inst["nb_vars"], inst["nb_lifted"], inst["Q_arr"]
    cs._load_neural_nets()
    sc = cs._gpu_new_scorer()
    sc.set_instance(inst["nb_vars"], inst["Q_arr"])
    n = sc.set_candidates_cover(inst["adj"], C3_DIM)
    assert n == int(g["nb_subproblems"])
    cs._agg_list = DeviceAgg(sc, n, inst["nb_vars"], inst["Q_arr"])
    cs._my_prob = harness.LinearRelaxation(np.zeros(inst["nb_lifted"] + inst["nb_vars"]))
    return cs, sc, pts, n


def c3_steps(cs, sc):
    """the three ways through one separation round: name -> f(strategy, LP point) -> number of cuts"""
    from sdpcutsel_via_nn_amd import harness

    def fused_rows(strat, vv):          # sdpcut_round_view: padded rows in the pinned block
        return int((sc.select_round(strat, SEL, copy=False, point=vv)["lam"] < -1e-15).sum())

    def fused_csr(strat, vv):           # sdpcut_round_csr: the cuts assembled on the device
        return sc.round_csr(strat, SEL, point=vv)["rhs"].shape[0]

    def dropin_pair(strat, vv):         # the reference's loop body, cut_select_qp.py:165-182, rows into a fresh row store
        cs._my_prob.linear_constraints = harness._RowStore()
        if strat == 4:
            _, picked = cs._sel_eigcut_by_ordering_on_measure(strat, vv, 1, sel_size=SEL)
        else:
            picked = cs._sel_eigcut_by_ordering_on_measure(strat, vv, 1)
        return cs._gen_eigcuts_selected(strat, SEL, picked, strong_only=False, vars_values=vv)

    return {"fused_rows": fused_rows, "fused_csr": fused_csr, "dropin_pair": dropin_pair}


def bench_c3(device_index, steps, warm=20):
    """ms per separation round on the c3 cover: {"strategy_4": {...}, "strategy_1": {...}} + score-kernel times"""
    import torch
    from sdpcutsel_via_nn_amd import _capi
    cs, sc, pts, n = c3_setup(device_index)
    out = {"cover": "%s dim %d, %d candidates of 2..%d variables (device enumeration)" % (C3_INSTANCE, C3_DIM, n, C3_DIM),
           "sel_size": SEL, "steps": steps,
           "points": "LP points of rounds %s of the reference trajectory (tests/golden)" % sorted(C3_ROUNDS.values())}
    fns = c3_steps(cs, sc)
    for strat in (4, 1):
        vv, res = pts[strat], {}
        for name, f in fns.items():
            for _ in range(warm):
                cuts = f(strat, vv)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                f(strat, vv)
            torch.cuda.synchronize()
            res[name + "_ms"] = (time.perf_counter() - t0) / steps * 1e3
            res["cuts"] = cuts
        # score kernels of this strategy's round (event pair on the dispatches: first to last size class)
        sc.set_option(_capi.OPT_TIMING, 1)
        ms = []
        for _ in range(10):
            fns["fused_csr"](strat, vv)
            ms.append(sc.last_timing()[0])
        sc.set_option(_capi.OPT_TIMING, 0)
        res["score_kernels_ms"] = float(np.mean(ms))
        if strat == 4:      # the MLP FLOPs of the cover's candidates over the scoring launch (every candidate of this cover has 4 variables)
            res["roofline_frac"] = FLOPS_PER_CAND[C3_DIM] * n / (res["score_kernels_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
        res["candidates_per_s_dropin"] = n / (res["dropin_pair_ms"] * 1e-3)
        out["strategy_%d" % strat] = res
    # (r4) the combined strategy at a GENERIC LP point as well (round 4 of the same trajectory): round 2 sits next to the McCormick
    # vertex, where a quarter of the lifted matrices have a reducible tridiagonal form and ~5 % a nearly multiple lambda_min -- the
    # worst case of csrc/lmin.h; rounds 4 and 5 are what a combined round costs once the cuts have moved the point
    g = np.load(os.path.join(ROOT, "tests", "golden", "rounds_%s_d%d_s4.npz" % (C3_INSTANCE.replace("-", "_"), C3_DIM)))
    if int(g["r04_strat"]) == 4:
        vv4, res4 = np.ascontiguousarray(g["r04_vars"], dtype=np.float64), {}
        for name in ("fused_csr", "dropin_pair"):
            for _ in range(warm):
                cuts = fns[name](4, vv4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fns[name](4, vv4)
            torch.cuda.synchronize()
            res4[name + "_ms"] = (time.perf_counter() - t0) / steps * 1e3
            res4["cuts"] = cuts
        sc.set_option(_capi.OPT_TIMING, 1)
        ms = []
        for _ in range(10):
            fns["fused_csr"](4, vv4)
            ms.append(sc.last_timing()[0])
        sc.set_option(_capi.OPT_TIMING, 0)
        res4["score_kernels_ms"] = float(np.mean(ms))
        res4["roofline_frac"] = FLOPS_PER_CAND[C3_DIM] * n / (res4["score_kernels_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
        out["strategy_4_round_4_generic_point"] = res4
    sc.close()
    return out


# ----------------------------------------------------------------------------- c5: the QCQP round (BASELINE configs[4]) on one GPU
C5_INSTANCE, C5_DIM = "q_50_10_25_1", 5


def bench_c5(device_index, steps, warm=10):
    """ms per QCQP separation round (cut_select_qcqp.py:64-98) on q_50_10_25_1 with 5-variable sub-problems: both covers and
    their intersection built on the device (4 + 1 377 077 candidates), the LP point of the round the reference itself ran
    (tests/golden/inst_qcqp50.npz), CutSolverQCQP.select_and_generate_round -- objective cover by the strategy, constraints-only
    cover by feasibility, (A + B)[0:5000], cuts handed to the LP's row store."""
    import torch
    from sdpcutsel_via_nn_amd import harness
    from sdpcutsel_via_nn_amd.cut_solver import CutSolverQCQP, DeviceAgg
    gold = os.path.join(ROOT, "tests", "golden")
    inst = harness.parse_osil(os.path.join(gold, "instances", C5_INSTANCE + ".osil"))
    vv = np.ascontiguousarray(np.load(os.path.join(gold, "inst_qcqp50.npz"))["vars"], dtype=np.float64)
    n = inst["nb_vars"]
    cs = CutSolverQCQP(device_index)
    cs._dim, cs._nb_vars, cs._nb_lifted, cs._Q_arr = C5_DIM, n, inst["nb_lifted"], inst["Q_arr"]
    cs._load_neural_nets()
    t0 = time.perf_counter()
    sc_o, sc_c = cs._gpu_new_scorer(), cs._gpu_new_scorer()
    for sc in (sc_o, sc_c):
        sc.set_instance(n, np.asarray(inst["Q_arr"], dtype=np.float64))
    n_o, n_c = sc_o.set_candidates_cover_split(sc_c, inst["adj"], inst["adj_cons"], C5_DIM)
    t_cover = time.perf_counter() - t0
    cover_obj, cover_cons = DeviceAgg(sc_o, n_o, n, inst["Q_arr"]), DeviceAgg(sc_c, n_c, n, inst["Q_arr"])
    cs._agg_list = cover_obj
    cs._my_prob = harness.LinearRelaxation(np.zeros(inst["nb_lifted"] + n))
    out = {"instance": "%s, %d-variable sub-problems: objective cover %d, constraints-only cover %d candidates (both built and "
                       "intersected on the device in %.0f ms incl. handle set-up)" % (C5_INSTANCE, C5_DIM, n_o, n_c, t_cover * 1e3),
           "sel_size": SEL, "steps": steps}
    for strat in (4, 1):
        def one():
            cs._my_prob.linear_constraints = harness._RowStore()
            return cs.select_and_generate_round(strat, vv, 1, SEL, cover_obj, cover_cons)
        for _ in range(warm):
            r = one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out["strategy_%d" % strat] = {"round_ms": ms, "cuts": r[2], "candidates_per_s": (n_o + n_c) / (ms * 1e-3)}
    # the same rounds with the two lists' rounds strictly one after the other (the pairing of DESIGN.md section 5 switched off)
    cs._gpu_overlap = False
    for b in cs._gpu_bindings.values():
        b.drain()
        b.follower = b.leader = None
    for strat in (4, 1):
        def one():
            cs._my_prob.linear_constraints = harness._RowStore()
            return cs.select_and_generate_round(strat, vv, 1, SEL, cover_obj, cover_cons)
        for _ in range(warm):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        out["strategy_%d" % strat]["round_ms_lists_one_after_the_other"] = (time.perf_counter() - t0) / steps * 1e3
    # speculative rounds of the follower pairing that nobody collected (cut_solver._Binding.drain); 0 in a loop that asks for both lists
    out["wasted_speculative_rounds"] = int(sum(getattr(b, "wasted", 0) for b in cs._gpu_bindings.values()))
    sc_o.close()
    sc_c.close()
    return out


MIXED_GOLDENS = ("rounds_spar100_050_1_d5_s4.npz", "rounds_spar070_050_1_d5_s4.npz")


def bench_mixed_cover(device_index, steps):
    """us per fused round (sdpcut_round_csr) on real covers with several size classes, shorter than the device is wide --
    the paper's dim-5 runs (generate_figs_tables.py:529-534): spar100-050-1 (72 673 five-, 103 four-, 1 three-variable sets) and
    spar070-050-1 (10 777 sets), enumerated on the device, at the LP points the reference's own trajectories recorded for round 2
    (combined strategy) and round 9 (feasibility); with the small classes on side streams and one launch after the other."""
    import torch
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd import _capi, harness
    import gc
    gold = os.path.join(ROOT, "tests", "golden")
    out = {}
    gc.collect()
    gc.freeze()      # (a generation-2 collection inside one of the short timed loops below would be most of its time)
    for fn in MIXED_GOLDENS:
        g = np.load(os.path.join(gold, fn))
        name, dim, sel = str(g["name"]), int(g["dim"]), int(g["sel_size"])
        inst = harness.parse_boxqp(os.path.join(gold, "instances", name + ".in"))
        sc = pkg.Scorer(device_index)
        sc.set_builtin_networks(dim)
        sc.set_instance(inst["nb_vars"], inst["Q_arr"])
        N = sc.set_candidates_cover(inst["adj"], dim)
        _, ks = sc.get_candidates(np.arange(N))
        rec = {"candidates": N, "sizes_2_to_5": np.bincount(ks, minlength=6)[2:].tolist(), "sel_size": sel, "steps": steps}
        for r in (2, 9):
            vv, strat = np.ascontiguousarray(g["r%02d_vars" % r]), int(g["r%02d_strat" % r])
            for one, side in ((1, 2), (0, 2), (0, 0)):      # the default: ONE launch over all size classes; a launch per class with the
                # small ones on side streams if that measures faster in this process; one launch after the other
                sc.set_option(_capi.OPT_ONE_LAUNCH, one)
                sc.set_option(_capi.OPT_SIDE_STREAMS, side)
                for _ in range(30):
                    sc.round_csr(strat, sel, point=vv)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    res = sc.round_csr(strat, sel, point=vv)
                us = (time.perf_counter() - t0) / steps * 1e6
                form = "" if one else ("_launch_per_class_side_streams_if_faster" if side else "_launch_per_class_one_after_the_other")
                rec["round_%d_strategy_%d%s" % (r, strat, form)] = {"round_us": us, "cuts": int(res["rhs"].shape[0])}
        sc.close()
        out[name + "_dim%d" % dim] = rec
    return out


def bench_triangle(device_index, steps):
    """ms per triangle-inequality separation step (cut_select_qp.py:824-863: all 4 T inequalities of spar125-075-1 evaluated and
    ranked by (density, violation) on the device, the selected ones handed to the LP's row store as one CSR block) at the LP
    point the reference's trajectory recorded for round 3."""
    from sdpcutsel_via_nn_amd import harness
    from sdpcutsel_via_nn_amd.cut_solver import CutSolver
    gold = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(gold, "rounds_spar125_075_1_d4_s4.npz"))
    inst = harness.parse_boxqp(os.path.join(gold, "instances", C3_INSTANCE + ".in"))
    n, L = inst["nb_vars"], inst["nb_lifted"]
    cs = CutSolver(device_index)
    cs._dim, cs._nb_vars, cs._nb_lifted, cs._Q_arr, cs._Q_adj = 4, n, L, inst["Q_arr"], inst["adj"]
    cs._my_prob = harness.LinearRelaxation(np.zeros(L + n))
    cs._preprocess_triangle_ineq()
    vv = g["r03_vars"]
    for _ in range(5):
        cs._my_prob.linear_constraints = harness._RowStore()
        nb = cs._separate_and_add_triangle(0.1, vv)
    t0 = time.perf_counter()
    for _ in range(steps):
        cs._my_prob.linear_constraints = harness._RowStore()
        nb = cs._separate_and_add_triangle(0.1, vv)
    dt = (time.perf_counter() - t0) / steps
    cs._gpu_tri.close()
    return {"instance": C3_INSTANCE, "triples": int(len(cs._gpu_tri_triples)), "inequalities": 4 * int(len(cs._gpu_tri_triples)),
            "cuts": int(nb), "step_ms": dt * 1e3, "steps": steps}


def gpu_round_record(sc, res, strat):
    """what selection_parity compares: the head of one GPU round (detached from the handle's pinned block) and the device's
    score arrays of that round (only the measures the strategy computes)"""
    eig, obj = sc.get_scores(eig=strat != 2, obj=strat != 1)
    return dict(idx=np.array(res["idx"]), score=np.array(res["score"]), new_strat=int(res["new_strat"]), eig=eig, obj=obj)


def bench_eig_only(make_scorer, K, n_local, vv_host, steps, gpu_rounds=None):
    """the feasibility round (strategy 1) on the main workload: eigenvalue-only kernel + selection + rows"""
    import torch
    from sdpcutsel_via_nn_amd import _capi
    sc, _, _, _ = make_scorer(K, n_local, 7, 0)
    for _ in range(30):
        sc.select_round(1, SEL, copy=False, point=vv_host)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sc.select_round(1, SEL, copy=False, point=vv_host)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sc.set_option(_capi.OPT_TIMING, 1)
    ms = []
    for _ in range(10):
        res = sc.select_round(1, SEL, copy=False, point=vv_host)
        ms.append(sc.last_timing()[0])
    if gpu_rounds is not None:
        gpu_rounds[1] = gpu_round_record(sc, res, 1)
    sc.close()
    k_ms = float(np.mean(ms))
    bytes_per = 4 * K + 8             # index set in, one fp64 out
    gbs = bytes_per * n_local / (k_ms * 1e-3) / 1e9
    return {"value": n_local / dt, "unit": "candidates/s", "ms_per_step": dt * 1e3, "steps": steps, "strategy": 1,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": None,
                         "kernel": "eig_only_kernel<%d, true>" % K,      # <largest size class present, counts the selection's leading digit>
                         "kernel_ms": k_ms, "candidates_per_launch": n_local, "bytes_per_candidate": bytes_per,
                         "issue": issue_floor(K, n_local, k_ms, "eig"),
                         "note": "gather + lambda_min in registers (Householder + Laguerre, csrc/lmin.h), no MLP: bound by VALU instruction "
                                 "issue (DESIGN.md section 5), the HBM figure is the algorithmic one"}}


def bench_opt_only(make_scorer, K, n_local, vv_host, steps, gpu_rounds=None):
    """the optimality round (strategy 2, cut_select_qp.py:569-601: the paper's choice for dense instances) on the main workload: the MLP
    without the eigenvalue, ranking by obj_improve, cut rows of the head (their lambda_min and eigenvector come from the epilogue)"""
    import torch
    from sdpcutsel_via_nn_amd import _capi
    sc, _, _, _ = make_scorer(K, n_local, 7, 0)
    for _ in range(30):
        sc.select_round(2, SEL, copy=False, point=vv_host)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sc.select_round(2, SEL, copy=False, point=vv_host)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sc.set_option(_capi.OPT_TIMING, 1)
    ms = []
    for _ in range(10):
        res = sc.select_round(2, SEL, copy=False, point=vv_host)
        ms.append(sc.last_timing()[0])
    if gpu_rounds is not None:
        gpu_rounds[2] = gpu_round_record(sc, res, 2)
    sc.close()
    k_ms = float(np.mean(ms))
    tflops = FLOPS_PER_CAND[K] * n_local / (k_ms * 1e-3) / 1e12
    return {"value": n_local / dt, "unit": "candidates/s", "ms_per_step": dt * 1e3, "steps": steps, "strategy": 2,
            "kernel_ms": k_ms, "achieved_TFLOPs": tflops, "roofline_frac": tflops / FP64_PEAK_TFLOPS}


def bench_cold_round(make_scorer, K, n_local, vv_host, repeats=20):
    """What a round costs when the device has been IDLE -- in the reference's loop a separation round follows an LP solve of
    0.1-10 s (cut_select_qp.py:149-200), not the previous round.  Median host-to-host time of one combined and one feasibility
    round on the main workload after the process slept 10 ms / 100 ms / 1 s (no GPU work in between), `repeats` times each, next
    to the same round issued back to back; and the same with a cheap mitigation measured: sdpcut_wake -- an empty kernel --
    issued when the sleep ends, i.e. while a caller would still be copying the LP solution (`_poked`)."""
    import torch
    sc, _, _, _ = make_scorer(K, n_local, 7, 0)
    out = {"repeats": repeats, "unit": "ms host to host, median", "workload": "the main list, sel_size %d" % SEL}
    for strat, name in ((4, "combined"), (1, "feasibility")):
        for _ in range(30):
            sc.select_round(strat, SEL, copy=False, point=vv_host)
        torch.cuda.synchronize()
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            sc.select_round(strat, SEL, copy=False, point=vv_host)
            ts.append(time.perf_counter() - t0)
        rec = {"back_to_back": float(np.median(ts)) * 1e3}
        for idle in (0.01, 0.1, 1.0):
            for poked in (False, True):
                ts = []
                for _ in range(repeats):
                    time.sleep(idle)
                    if poked:      # sdpcut_wake as soon as the LP solve returns; the round follows ~20 us later (the point's copy)
                        sc.wake()
                        t_p = time.perf_counter()
                        while time.perf_counter() - t_p < 20e-6:
                            pass
                    t0 = time.perf_counter()
                    sc.select_round(strat, SEL, copy=False, point=vv_host)
                    ts.append(time.perf_counter() - t0)
                rec["idle_%g_ms%s" % (idle * 1e3, "_poked" if poked else "")] = float(np.median(ts)) * 1e3
        out[name] = rec
    sc.close()
    return out


class _StdoutToStderr(object):
    """fd-level redirect: native libraries (RCCL prints a version banner when its first communicator
    comes up) must not put lines on stdout, which carries exactly one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


# Instruction-issue floor of the two hot kernels at k = 3, from the PMC passes kept under profiles/ (instructions per launch of
# 10^6 candidates; a VALU instruction occupies its SIMD for 4 cycles, a quarter-rate transcendental for 16, an fp64 MFMA for 64):
# what the kernel would take if its SIMDs never waited.  Not a roofline in the HBM / MFMA sense -- the bound these kernels run into.
ISSUE_CYCLES_PER_CAND = {
    # (r4, Householder + Laguerre lambda_min) VALU instructions per launch, of which quarter-rate transcendentals, MFMAs:
    ("eig", 3): ((8.065 - 0.229) * 4 + 0.229 * 16, "profiles/r04_eig_k3_kernel_pmc.txt (8.065 M VALU of which 0.229 M v_rsq / v_rcp_f64; round 3, Jacobi: 18.67 M)"),
    ("mfma", 3): ((73.737 - 1.015) * 4 + 1.015 * 16 + 5.4375 * 64, "profiles/r05_k3_score_kernel_pmc.txt (73.74 M VALU of which 1.015 M transcendental + 5.4375 M MFMA "
                                                                   "per launch = 4719 + 348 per 64 candidates, the partial strips of the balanced last round and the "
                                                                   "fine histogram of the selection's class included; round 4: 72.31 M VALU, round 3: 83.71 M)"),
}
SIMDS, CLOCK_GHZ = 1024, 2.4


def issue_floor(k, n_per_launch, kernel_ms, which):
    e = ISSUE_CYCLES_PER_CAND.get((which, k))
    if e is None:
        return None
    floor_ms = e[0] * n_per_launch / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
    return {"floor_ms": floor_ms, "frac": floor_ms / kernel_ms, "clock_GHz": CLOCK_GHZ, "source": e[1]}


def roofline(k, n_per_launch, kernel_ms, traffic, counts_digit=True, variant="mfma"):
    tflops = FLOPS_PER_CAND[k] * n_per_launch / (kernel_ms * 1e-3) / 1e12
    gbs = BYTES_PER_CAND[k] * n_per_launch / (kernel_ms * 1e-3) / 1e9
    return {"bound": "mfma", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS,
            "traffic": traffic, "kernel": kernel_name(k, counts_digit and variant == "mfma", variant), "kernel_ms": kernel_ms, "candidates_per_launch": n_per_launch,
            "flops_per_candidate": FLOPS_PER_CAND[k], "hbm_algorithmic_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
            "bytes_per_candidate": BYTES_PER_CAND[k], "issue": issue_floor(k, n_per_launch, kernel_ms, variant)}


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset, N > 1): THIS process -- which has made no GPU call and
    never makes one -- builds the library once, starts N fresh children of this same script (one rank per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would), relays rank 0's one JSON line to stdout and
    returns the worst child's exit code.  Children are separate processes started with fork + exec from a GPU-free parent: no
    exec after a GPU call anywhere.  A rank that dies takes the others down with it (by pid) instead of leaving them in a
    collective."""
    import socket
    import subprocess
    # A profiler's preload (rocprofv3: LD_PRELOAD / ROCP* variables) initialises the GPU in THIS process before main() runs, and
    # starting the ranks from a process that holds a GPU is the exec this pool's machines refuse.  Profile one rank
    # (`rocprofv3 ... -- python3 bench.py --gpus 1`) or put the profiler inside the launch (`torch.distributed.run ... rocprofv3`).
    # (LD_PRELOAD alone says nothing: the pool's own boxes preload a bookkeeping hook into every process.  rocprofv3 announces
    # itself through ROCP_TOOL_LIBRARIES / ROCPROFILER_* and a preload of its tool library.)
    preload = [k for k in os.environ if k.startswith("ROCP_TOOL") or k.startswith("ROCPROFILER_")]
    if any(t in os.environ.get("LD_PRELOAD", "").lower() for t in ("rocprof", "roctracer")):
        preload.append("LD_PRELOAD=" + os.environ["LD_PRELOAD"])
    if preload:
        sys.stderr.write("bench.py: refusing to self-launch %d ranks under a profiler / preload (%s): the parent must not have touched "
                         "a GPU.  Profile a single rank, or launch with torch.distributed.run and profile inside it.\n"
                         % (n_ranks, ", ".join(sorted(preload))))
        return 2
    from sdpcutsel_via_nn_amd import build as hip_build
    hip_build.build(verbose=True)           # compile only: no dlopen, no GPU
    # rendezvous port: a bound (never listening) SO_REUSEADDR socket keeps the number ours until the ranks are gone -- rank 0's
    # store binds the same port with SO_REUSEADDR, which Linux allows next to a non-listening holder, while a stranger's plain
    # bind() is refused (the bind-close-reuse of round 4 left a window)
    holder = socket.socket()
    holder.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    holder.bind(("127.0.0.1", 0))
    port = holder.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   GROUP_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SDPCUT_BENCH_SELF_LAUNCHED="1")
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports only dmabuf IPC; with the runtime's default (legacy IPC
        # handles) RCCL's intra-node transport and any device-tensor sharing between processes fail at set-up with
        # `hipIpcGetMemHandle: invalid argument` (the pool's environment notes; the variable is already exported on its
        # machines -- setdefault keeps whatever the caller's environment says, this only covers a scrubbed environment).
        # It changes how buffers are SHARED between ranks, nothing a single rank computes.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        # rank 0's stdout is read here (exactly one JSON line goes on); every other rank's stdout joins stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(ln.decode(errors="replace") for ln in procs[0].stdout), daemon=True)
    reader.start()
    worst, live = 0, list(procs)
    deadline = None
    while live:
        time.sleep(0.05)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                if deadline is None:        # the others would wait for this rank in the next collective
                    deadline = time.monotonic() + 5.0
        if deadline is not None and live and time.monotonic() > deadline:
            for p in live:
                p.kill()
            deadline = float("inf")
    reader.join(timeout=10)
    holder.close()
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines[-1:]:
            sys.stderr.write(ln)
    if json_lines:
        sys.stdout.write(json_lines[-1] if json_lines[-1].endswith("\n") else json_lines[-1] + "\n")
        sys.stdout.flush()
    elif worst == 0:
        worst = 1
    return worst


# What a step should cost per rank at N = 8 (from the one-GPU phase measurements under profiles/): a SCALE
# line is to be held against this.  us per phase; "step_ms" is their sum (the host's enqueue calls overlap the score kernel).
EXPECTED_8GPU = {      # (r5: one source of truth -- DESIGN.md section 6 quotes THIS table; phases from profiles/r05_bench_forced_sharded.json, r05_bench_one_rank_rccl.json)
    "c2": {"score_and_head_us": 345, "all_gather_us": [20, 35], "merge_and_rows_us": [25, 30], "host_tail_us": 15,
           "step_ms": [0.40, 0.43], "aggregate_candidates_per_s": [1.86e10, 2.0e10], "weak_scaling_efficiency": [0.88, 0.94]},
    "c4": {"score_and_head_us": 4060, "selection_tail_point_and_collective_us": 100, "step_ms": [4.15, 4.25],
           "strong_scaling_efficiency_vs_one_gpu_step": 0.97},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS) + ["c3"], default="c2")
    ap.add_argument("--no-c3", action="store_true", help="skip secondary.c3 / c5 (the spar125-075-1 dim-4 rounds, the q_50 QCQP round) of the default run")
    ap.add_argument("--kernel", choices=["mfma", "simple", "valu"], default="mfma")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip secondary.cold_round (rounds after 10 ms / 100 ms / 1 s of idle device: ~50 s)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the k = 2, 4, 5 single-GPU rates")
    ap.add_argument("--device-point", action="store_true",
                    help="A/B: take the LP point from a device buffer (the round-1 bracket) instead of host memory")
    ap.add_argument("--no-fuse-keys", action="store_true",
                    help="A/B: the selection runs its own key pass instead of starting from the score kernel's histograms")
    ap.add_argument("--no-auto-regime", action="store_true",
                    help="A/B: combined strategy assumes the common regime, the host repeats the selection otherwise")
    ap.add_argument("--no-fused-tail", action="store_true", help="A/B: one launch per selection pass")
    ap.add_argument("--no-prefilter", action="store_true",
                    help="A/B: the score kernels do not count the fine histogram of the class; the selection runs its radix passes (rounds 2-4)")
    ap.add_argument("--no-pinned-point", action="store_true",
                    help="A/B (c4 configs): hand the LP point over in an ordinary host array instead of the handle's pinned buffer")
    ap.add_argument("--two-calls", action="store_true", help="A/B: sdpcut_set_point + sdpcut_select_round_view instead of sdpcut_round_view")
    ap.add_argument("--coop", action="store_true", help="A/B: cooperative launch of the fused selection kernel (+20 us per round)")
    ap.add_argument("--cpu-sample", type=int, default=10 ** 6)
    ap.add_argument("--k", type=int, choices=[2, 3, 4, 5], default=None, help="candidate size (default: the config's, 3)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="attach the HIP event pair to the score dispatch of every n-th step of the timed region")
    args = ap.parse_args()
    if args.config == "c3":
        return main_c3(args)
    cfg = CONFIGS[args.config]
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: be the launcher (before torch is imported, before anything touches a GPU)
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a SCALE line must mean what its --gpus says: refuse instead of running another size under that label
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if os.environ.get("SDPCUT_BENCH_LAUNCH_ONLY") == "1":
        # launcher rehearsal without a GPU (tests/test_bench_contract.py, CPU): rendezvous over gloo, one collective, one line
        if os.environ.get("SDPCUT_BENCH_FAIL_RANK") == str(rank):
            sys.exit(3)
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print("a stray line from a native library")
            print(json.dumps({"launch_only": True, "n_gpus": world, "rank_sum": int(t.item()), "steps": args.steps,
                              "self_launched": os.environ.get("SDPCUT_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        dist.destroy_process_group()
        return
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
    import __graft_entry__ as entry
    with _StdoutToStderr():
        if use_dist:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
        if rank == 0:
            entry.build()
        if use_dist:
            dist.barrier()          # the first collective: the communicator (and RCCL's banner) come up here

    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    from sdpcutsel_via_nn_amd.distributed import DeviceOps, ShardedSelector

    nb_vars, K = cfg["nb_vars"], args.k or cfg["k"]
    n_local = cfg["per_gpu"] if cfg["per_gpu"] else cfg["total"] // world
    kernel_opt = {"mfma": _capi.KERNEL_MFMA, "simple": _capi.KERNEL_SIMPLE, "valu": _capi.KERNEL_VALU}[args.kernel]

    def make_scorer(k, count, seed, base):
        """handle with the network of size k, the instance of (nb_vars, seed 7) and `count` candidates"""
        sc = _capi.Scorer(local_rank)
        if args.no_fuse_keys:
            sc.set_option(_capi.OPT_FUSE_KEYS, 0)
        if args.no_auto_regime:
            sc.set_option(_capi.OPT_AUTO_REGIME, 0)
        if args.no_fused_tail:
            sc.set_option(_capi.OPT_FUSED_TAIL, 0)
        if args.coop:
            sc.set_option(_capi.OPT_COOP_LAUNCH, 1)
        if args.no_prefilter:
            sc.set_option(_capi.OPT_PREFILTER, 0)
        sc.set_option(_capi.OPT_KERNEL, kernel_opt)
        sc.set_network(k, *networks.load_network(k))
        Q_arr, vv, _ = synthetic.make_instance(nb_vars, seed=7)      # one LP point and one objective for the whole job
        sc.set_instance(nb_vars, Q_arr)
        sets = None
        if args.config == "c2":
            sets = synthetic.make_workload(nb_vars=nb_vars, k=k, count=count, seed=seed)["set_inds"]
            sc.set_candidates(sets, np.full(count, k, dtype=np.int32), global_base=base)
        else:
            sc.set_candidates_philox(k, count, seed=7, first_id=base)    # index sets never exist on the host
        return sc, Q_arr, vv, sets

    sc, Q_arr, vv_host, sets_host = make_scorer(K, n_local, 7 + rank, rank * n_local)
    pinned_point = args.config != "c2" and not args.no_pinned_point and not args.device_point
    if pinned_point:
        # n = 1000: the LP point is 4 MB.  A caller that lets its LP solver write the solution into the handle's pinned
        # staging block (sdpcut_point_buffer) saves the library's host copy of it (~110 us per round)
        vv_pageable = vv_host
        vv_host = sc.point_buffer()
        vv_host[:] = vv_pageable
    d_vars = torch.from_numpy(vv_host).to(device) if args.device_point else None
    sel = None
    if world > 1 or force_sharded:
        # DeviceOps binds the library to torch's current stream: its kernels, torch's copies and the
        # hand-off to the collective are ordered without host synchronisation (a dedicated
        # non-blocking stream measured no better at N = 1 and worse with two ranks on one GPU)
        sel = ShardedSelector(DeviceOps(sc, device), n_local)

    def make_step(sc, sel, kernel_ms):
        """One round.  The score kernel's duration is measured live with a HIP event pair attached to its
        dispatch (hipExtLaunchKernelGGL) on every `--time-every`-th step: the pair and its read-back cost the
        host ~10 us, which a production round does not pay, so it rides on a sample of the timed steps."""
        counter = [0]

        def step():
            timed = counter[0] % args.time_every == 0
            counter[0] += 1
            if timed:
                sc.set_option(_capi.OPT_TIMING, 1)
            if sel is None and d_vars is None and not args.two_calls:
                # one C-ABI call: LP point host -> device (41 KB at n = 100, 4 MB at n = 1000; part of every real
                # round) -> score (eig + NN) -> combined ranking -> cut rows of the head -> results written by the
                # device into the handle's pinned host block (copy=False hands out views of it)
                res = sc.select_round(4, SEL, copy=False, point=vv_host)
            elif sel is None:
                if d_vars is not None:
                    sc.set_point_device(d_vars.data_ptr())
                else:
                    sc.set_point(vv_host)
                res = sc.select_round(4, SEL, copy=False)
            else:
                sc.set_point(vv_host)
                # score the shard + packed head record (one call) -> ONE all-gather (RCCL) -> replicated merge ->
                # each rank generates the rows of its own candidates -> one D2H, one host sync
                res = sel.select_round(4, SEL, copy=False)
            if timed:
                kernel_ms.append(sc.last_timing()[0])
                sc.set_option(_capi.OPT_TIMING, 0)
            return res
        step.counter = counter
        return step

    kernel_ms = []
    step = make_step(sc, sel, kernel_ms)

    # Setup, untimed: the GPU comes out of idle with low clocks and needs ~50 ms of load to reach the
    # sustained state (score kernel 0.42 -> 0.39 ms); bring it there before the W warmup steps so that
    # short runs (K = 20) measure the same machine state as long ones.
    # A full (generation-2) Python garbage collection walks the millions of objects `import torch`
    # creates and stalls the host for ~75 ms once every few hundred steps: park them in the permanent
    # generation, as latency-sensitive Python services do.
    import gc
    gc.collect()
    gc.freeze()
    prewarm = max(3, int(PREWARM_STEPS * min(1.0, 10 ** 6 / n_local)))
    for _ in range(prewarm):
        step()
    for _ in range(args.warmup):
        step()
    del kernel_ms[:]
    step.counter[0] = 0          # the first step of the timed region carries an event pair
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_res = step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0      # this rank's K steps, before it waits for the others
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    # the LAST round of the timed region, kept for the parity object of the line (compared with the ranking of the oracle's own
    # scores after everything is timed, cpu_baseline): ids / scores as returned, the device's score arrays of that round
    gpu_rounds = {}
    if world == 1 and sel is None and not args.no_cpu_baseline:
        gpu_rounds[4] = gpu_round_record(sc, last_res, 4)
    rank_ms = None
    if use_dist:
        dev_t = device if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # stragglers: every rank's own time for its K steps, before it waits for the others (the line's value uses the MAX of the
        # barrier-to-barrier time)
        mine = torch.tensor([dt_own], dtype=torch.float64, device=dev_t)
        every = [torch.zeros(1, dtype=torch.float64, device=dev_t) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        dt = float(t.item())

    phases = None
    if sel is not None:
        # Where a sharded step goes (every rank runs the loop -- it contains the collective --, rank 0 reports): device
        # time between events on the stream the library and the collective share, host time around each call.  Untimed
        # extra steps after the measured region; the events cost the host a few microseconds each.
        from sdpcutsel_via_nn_amd import _capi as capi
        ops = sel.ops
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = np.zeros(8)
        n_ph = 30
        for _ in range(n_ph):
            t0 = time.perf_counter()
            sc.set_point(vv_host)
            t1 = time.perf_counter()
            ev[0].record()
            rec = ops.shard_head(capi.PART_STRONG, SEL)
            ev[1].record()
            t2 = time.perf_counter()
            allrec = sel._all_gather(rec)
            ev[2].record()
            t3 = time.perf_counter()
            ops.shard_finish_enqueue(world, SEL, allrec, SEL)
            ev[3].record()
            t4 = time.perf_counter()
            ops.shard_finish_wait()
            t5 = time.perf_counter()
            torch.cuda.synchronize()
            acc += (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, ev[0].elapsed_time(ev[1]) * 1e-3, ev[1].elapsed_time(ev[2]) * 1e-3,
                    ev[2].elapsed_time(ev[3]) * 1e-3)
        acc *= 1e6 / n_ph
        phases = {"steps": n_ph, "rank": 0,
                  "host_us": {"set_point": acc[0], "shard_head_enqueue": acc[1], "all_gather_enqueue": acc[2], "finish_enqueue": acc[3],
                              "finish_wait": acc[4]},
                  "device_us": {"score_and_head": acc[5], "all_gather": acc[6], "merge_and_rows": acc[7]},
                  "note": "score_and_head = point copy + score kernel + per-shard top-k into the packed record; all_gather = %d records of "
                          "%d bytes (%s); merge_and_rows = replicated merge + eigen-cut rows of the own entries written to pinned host "
                          "memory; finish_wait = the round's one host wait" % (world, (8 + 2 * SEL) * 8, backend if use_dist else "no collective")}
    if rank == 0:
        total = n_local * world * args.steps
        k_ms = float(np.mean(kernel_ms))
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "score_kernel_traffic.json")
        if os.path.exists(tfile) and args.config == "c2" and K == 3:
            # HBM bytes per 1e6-candidate launch of this kernel from the PMC passes kept under profiles/
            # (separate rocprofv3 --pmc runs, gfx950 corrections); a profile figure, not measured in this run
            traffic = json.load(open(tfile)).get(args.kernel)
        out = {
            "metric": BASELINE_METRIC, "value": total / dt, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["text"], "config": args.config, "candidates_per_gpu": n_local, "nb_vars": nb_vars, "k": K,
                       "sel_size": SEL, "kernel": args.kernel, "strategy": 4,
                       "bracket": "device point -> host results" if args.device_point else
                                  ("LP point in the handle's pinned buffer (sdpcut_point_buffer) -> host results" if pinned_point
                                   else "host point -> host results")},
            "roofline": roofline(K, n_local, k_ms, traffic, counts_digit=not args.no_fuse_keys, variant=args.kernel),
        }
        if phases is not None:
            out["phases"] = phases
        if use_dist:
            out["config"]["collective_backend"] = backend
            out["config"]["rccl_ranks"] = dist.get_world_size() if backend == "nccl" else 0      # ranks of the RCCL communicator
            out["ms_per_step_ranks"] = {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms,
                                        "note": "each rank's own K steps before the closing barrier; ms_per_step is the barrier-to-barrier MAX"}
            out["config"]["ms_per_step_rank_min"], out["config"]["ms_per_step_rank_max"] = min(rank_ms), max(rank_ms)
            out["config"]["launcher"] = "self (python bench.py --gpus N)" if os.environ.get("SDPCUT_BENCH_SELF_LAUNCHED") == "1" \
                else "torch.distributed.run"
            out["config"]["parallelism"] = "candidate shards x%d, one all-gather of per-shard heads per round" % world
        if world > 1 and args.config in EXPECTED_8GPU:
            out["expected_8gpu"] = dict(EXPECTED_8GPU[args.config], source="DESIGN.md section 6 (one-GPU phase measurements)")
        out["roofline"]["kernel_ms_samples"] = len(kernel_ms)
        out["config"]["selection_fallbacks"] = sc.get_stat(_capi.STAT_SELECT_FALLBACKS)       # rounds answered by the full-sort path
        out["roofline"]["traffic_source"] = ("profiles/score_kernel_traffic.json (rocprofv3 --pmc passes of this kernel at "
                                             "this size; not re-measured in this run)") if traffic else None
        if world == 1 and args.config == "c2" and not args.no_secondary:
            # SURVEY 8 d: "+ k = 2, 4, 5 single GPU as secondary" -- same round, same sizes, the other three networks
            sec = {}
            for k2 in (2, 4, 5):
                s2, _, _, _ = make_scorer(k2, n_local, 7 + k2, 0)
                ms2 = []
                st2 = make_step(s2, None, ms2)
                for _ in range(30):
                    st2()
                del ms2[:]
                st2.counter[0] = 0
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n2 = max(20, args.steps // 4)
                for _ in range(n2):
                    st2()
                torch.cuda.synchronize()
                d2 = time.perf_counter() - t1
                r2 = roofline(k2, n_local, float(np.mean(ms2)), None, counts_digit=not args.no_fuse_keys, variant=args.kernel)
                sec["k%d" % k2] = {"value": n_local * n2 / d2, "unit": "candidates/s", "ms_per_step": d2 / n2 * 1e3, "steps": n2,
                                   "kernel_ms": r2["kernel_ms"], "roofline_frac": r2["frac"], "achieved_TFLOPs": r2["achieved"]}
                s2.close()
            out["secondary"] = sec
            for k2 in (2, 4, 5):
                out["roofline"]["k%d_frac" % k2] = sec["k%d" % k2]["roofline_frac"]
            # the feasibility round (strategy 1, cut_select_qp.py:639-654) on the same list: the eigenvalue-only kernel
            out["eig_only"] = bench_eig_only(make_scorer, K, n_local, vv_host, max(20, args.steps // 4), gpu_rounds)
            out["roofline"]["eig_only_ms_per_step"] = out["eig_only"]["ms_per_step"]
            out["secondary"]["strategy_2"] = bench_opt_only(make_scorer, K, n_local, vv_host, max(20, args.steps // 4), gpu_rounds)
            if not args.no_cold:
                out["secondary"]["cold_round"] = bench_cold_round(make_scorer, K, n_local, vv_host)
            if not args.no_c3:
                out["secondary"]["c3"] = bench_c3(local_rank, max(20, args.steps // 4))
                # (scalars inside `roofline` survive the driver's record of the line: the lowest scoring kernel and the real-cover figures)
                out["roofline"]["c3_round2_frac"] = out["secondary"]["c3"]["strategy_4"]["roofline_frac"]
                out["roofline"]["c3_round4_frac"] = out["secondary"]["c3"].get("strategy_4_round_4_generic_point", {}).get("roofline_frac")
                out["secondary"]["c5"] = bench_c5(local_rank, max(20, args.steps // 4))
                out["secondary"]["mixed_cover"] = bench_mixed_cover(local_rank, max(50, args.steps // 2))
                out["secondary"]["triangle"] = bench_triangle(local_rank, max(20, args.steps // 4))
        if world == 1 and not args.no_cpu_baseline:
            vv_host = np.array(vv_host)     # (detach from the handle's pinned buffer: worker processes pickle it)
            if sets_host is None:       # device-generated list: the numpy twin of the generator names the sample
                m = min(args.cpu_sample, n_local)
                sets_host = synthetic.philox_index_sets(nb_vars, K, np.arange(m), seed=7)
            out["cpu_baseline"] = cpu_baseline(sets_host, nb_vars, K, vv_host, Q_arr, n_local, min(args.cpu_sample, n_local), gpu_rounds)
            par = out["cpu_baseline"].pop("parity", None)
            if par is not None:
                # the GPU rounds of THIS run against the ranking of the oracle's own scores on the whole list (cut_select_qp.py:601-632)
                out["parity"] = par
                for key in ("topk_identical", "positions_differing", "ids_one_side_only", "max_abs_d_eig", "max_rel_d_obj"):
                    out["cpu_baseline"]["parity_" + key] = par.get(key)      # (scalars survive the driver's record of the line)
                for s_ in (1, 2):
                    if "strategy_%d" % s_ in par:
                        out["cpu_baseline"]["parity_strategy_%d_topk_identical" % s_] = par["strategy_%d" % s_]["topk_identical"]
        print(json.dumps(out), flush=True)
    sc.close()
    bad_comm = False
    if use_dist:
        # the multi-GPU line is a statement about RCCL over xGMI: a run whose ranks did not all join ONE RCCL communicator of
        # --gpus ranks must not pass as one -- the line above is printed, the exit code says no (the one-GPU rehearsal forms,
        # which say what they are in config.collective_backend / rccl_ranks, are exempt)
        rehearsal = backend != "nccl" or os.environ.get("SDPCUT_BENCH_ONE_DEVICE") == "1" or solo_dist
        bad_comm = not rehearsal and (dist.get_backend() != "nccl" or dist.get_world_size() != args.gpus)
        dist.destroy_process_group()
    if bad_comm:
        sys.exit("bench.py: the RCCL communicator has %d ranks, --gpus asked for %d" % (world, args.gpus))


def main_c3(args):
    if args.no_prefilter:      # A/B: every handle the drop-in classes make runs without the fine histogram (same results either way)
        from sdpcutsel_via_nn_amd import _capi
        plain_init = _capi.Scorer.__init__

        def init_without(self, *a, **kw):
            plain_init(self, *a, **kw)
            self.set_option(_capi.OPT_PREFILTER, 0)
        _capi.Scorer.__init__ = init_without
    """--config c3: the spar125-075-1 dim-4 rounds as the line's workload (one GPU).  value = candidates per second
    through the drop-in pair with the combined strategy (every candidate scored eig + NN, the round ranked and its
    cuts handed to the LP's row store); the other routes and the feasibility point ride along in `c3`."""
    import torch
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        sys.exit("--config c3 is a one-GPU workload (a real cover does not grow with the number of GPUs)")
    torch.cuda.set_device(0)
    import __graft_entry__ as entry
    with _StdoutToStderr():
        entry.build()
    import gc
    gc.collect()
    gc.freeze()
    res = bench_c3(0, args.steps, warm=max(args.warmup, 20))
    n = int(res["cover"].split(",")[1].split()[0])
    s4 = res["strategy_4"]
    flops = 11500.0        # per candidate of the dominant class (4-variable sets); the event pair spans all three size classes
    tflops = flops * n / (s4["score_kernels_ms"] * 1e-3) / 1e12
    out = {"metric": BASELINE_METRIC, "value": n / (s4["dropin_pair_ms"] * 1e-3), "unit": "candidates/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": s4["dropin_pair_ms"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "instance file + recorded LP points (tests/golden)",
           "config": {"workload": "configs[2]: " + res["cover"] + ", combined strategy, sel_size 5000, through the drop-in pair "
                                  "_sel_eigcut_by_ordering_on_measure + _gen_eigcuts_selected", "config": "c3"},
           "roofline": {"bound": "mfma", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS,
                        "traffic": None, "kernel": "score_mfma_kernel<2|3|4, ...> (three size classes, first start to last end)",
                        "kernel_ms": s4["score_kernels_ms"], "candidates_per_launch": n, "flops_per_candidate": flops},
           "c3": res}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

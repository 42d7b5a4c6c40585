"""Two ranks sharing the one GPU of the test box (gloo rendezvous, host-staged all-gather):
the sharded selection must reproduce the single-list ranking head exactly."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from sdpcutsel_via_nn_amd import _capi, networks, synthetic
from sdpcutsel_via_nn_amd.distributed import DeviceOps, ShardedSelector
from oracle import cutsel_oracle as oracle

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
n_each, nv = 60000, 60
wl = synthetic.make_workload(nb_vars=nv, k=3, count=n_each * world, seed=11)
lo = rank * n_each
sc = _capi.Scorer(0)
sc.set_network(3, *networks.load_network(3))
sc.set_instance(nv, wl["Q_arr"])
sc.set_candidates(wl["set_inds"][lo:lo + n_each], wl["ks"][lo:lo + n_each], global_base=lo)
points = [wl["vars_values"]]
x = np.full(nv, 0.5); iu = np.triu_indices(nv)
rng = np.random.default_rng(2)
points.append(np.concatenate([np.where(rng.uniform(size=iu[0].shape[0]) < 0.5, 0.25, 0.5), x]))   # few strong candidates
sel = ShardedSelector(DeviceOps(sc, dev), n_each)
for vv in points:
    sc.set_point(vv)
    sc.score(_capi.EIG | _capi.NN)
    eig, obj = sc.get_scores()
    parts_e = [torch.empty(n_each, dtype=torch.float64) for _ in range(world)]
    parts_o = [torch.empty(n_each, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts_e, torch.from_numpy(eig))
    dist.all_gather(parts_o, torch.from_numpy(obj))
    E, O = torch.cat(parts_e).numpy(), torch.cat(parts_o).numpy()
    n_strong = int(((O > 0) & (E < -1e-15)).sum())
    for strat in (1, 2, 4):
        for sel_size in sorted({37, 5000, 12000, max(n_strong - 5, 1), n_strong + 40}):
            sel.ops.ensure_scored(strat)
            r = sel.select(strat, sel_size)
            order, score, new_strat, cnt = oracle.rank_arrays(strat, O, E, sel_size)
            k = min(sel_size, order.shape[0])
            assert np.array_equal(r["ids"].cpu().numpy(), order[:k]), (strat, sel_size, rank)
            assert np.array_equal(r["scores"].cpu().numpy(), score[:k] + 0.0), (strat, sel_size)
            assert r["new_strat"] == new_strat
            if strat == 4:
                assert r["counters"]["strong"] == cnt["strong"] and r["counters"]["violated"] == cnt["violated"]
            # fused round (shard_head / all-gather / shard_finish): same head + rows of the own entries
            if sel_size != 37:
                sc.set_point(vv)       # scores gone: the round scores the shard itself (leading digit counted by the score kernels)
            q = sel.select_round(strat, sel_size)
            assert np.array_equal(q["ids"], order[:k]), (strat, sel_size, rank)
            assert np.array_equal(q["scores"], score[:k] + 0.0), (strat, sel_size)
            assert q["new_strat"] == new_strat
            if strat == 4:
                assert q["counters"]["strong"] == cnt["strong"] and q["counters"]["violated"] == cnt["violated"]
            own = (order[:k] >= lo) & (order[:k] < lo + n_each)
            assert np.array_equal(q["mine"], own), (strat, sel_size)
            lam, coef, rhs, _, ks = sc.cut_rows(order[:k][own] - lo)
            assert np.array_equal(q["lam"], lam) and np.array_equal(q["rhs"], rhs) and np.array_equal(q["ks"], ks)
            assert np.array_equal(q["coef"], coef[:, :9])
            assert np.allclose(q["lam"], E[order[:k][own]], rtol=1e-9, atol=1e-12)
# every fused round was served inside the library: one collective (common regime) or two (every entry visited)
assert sel.path_counts["unfused"] == 0 and sel.path_counts["comball"] >= 2 and sel.path_counts["common"] >= 10, sel.path_counts
sc.close()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_shard_round_ragged_shards_one_process(oracle):
    """sdpcut_shard_head_device / sdpcut_shard_finish_round with four handles standing in for four
    ranks (shards of 20000, 0, 7 and 12000 candidates; the records are concatenated instead of
    all-gathered): merged head = the single-list ranking head, rows = each shard's own rows."""
    import numpy as np
    import torch
    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    from sdpcutsel_via_nn_amd.distributed import DeviceOps

    dev = torch.device("cuda", 0)
    sizes = [20000, 0, 7, 12000]
    bases = np.concatenate([[0], np.cumsum(sizes)])
    nv = 40
    wl = synthetic.make_workload(nb_vars=nv, k=3, count=int(bases[-1]), seed=23)
    opss = []
    for r, n in enumerate(sizes):
        sc = _capi.Scorer(0)
        sc.set_network(3, *networks.load_network(3))
        sc.set_instance(nv, wl["Q_arr"])
        lo = int(bases[r])
        sc.set_candidates(wl["set_inds"][lo:lo + n], wl["ks"][lo:lo + n], global_base=lo)
        opss.append(DeviceOps(sc, dev))
    try:
        E, O = [], []
        for ops in opss:
            ops.scorer.set_point(wl["vars_values"])
            ops.scorer.score(_capi.EIG | _capi.NN)
            e, o = ops.scorer.get_scores()
            E.append(e[:ops.scorer.N]); O.append(o[:ops.scorer.N])
        E, O = np.concatenate(E), np.concatenate(O)
        n_strong = int(((O > 0) & (E < -1e-15)).sum())
        assert n_strong > 50
        for strat, code in ((1, 1), (2, 2), (4, _capi.PART_STRONG)):
            for sel in (1, 29, 50, 5000, 8192, 16384):
                if strat == 4 and sel > n_strong:
                    continue                       # the general regime is ShardedSelector.select's business
                if sel in (29, 5000):             # scores gone: shard_head scores the shard itself
                    for ops in opss:
                        ops.scorer.set_point(wl["vars_values"])
                allrec = torch.cat([ops.shard_head(code, sel) for ops in opss])
                order, score, _, _ = oracle.rank_arrays(strat, O, E, sel)
                k = min(sel, order.shape[0])
                for r, ops in enumerate(opss):
                    out = ops.shard_finish(len(sizes), sel, allrec, sel)
                    g = out["headers"].sum(axis=0)
                    assert int(g[0]) == (order.shape[0] if strat != 4 else n_strong)
                    # (a head that scores the shard itself computes only the measure its strategy ranks by:
                    # the other counter covers the shards that happen to hold that measure already)
                    n_viol, n_pos = int((E < -1e-15).sum()), int((O > 0).sum())
                    assert int(g[1]) == n_viol if strat != 2 else 0 <= int(g[1]) <= n_viol
                    assert int(g[2]) == n_pos if strat != 1 else 0 <= int(g[2]) <= n_pos
                    assert np.array_equal(out["idx"][:k], order[:k]), (strat, sel, r)
                    assert np.array_equal(out["score"][:k] + (1000.0 if strat == 4 else 0.0), score[:k] + 0.0)
                    assert np.all(out["idx"][k:] == np.iinfo(np.int64).max) and np.all(out["ks"][k:] == 0)
                    lo, n = int(bases[r]), sizes[r]
                    own = (order[:k] >= lo) & (order[:k] < lo + n)
                    assert np.array_equal(out["ks"][:k] > 0, own)
                    assert np.all(np.isnan(out["lam"][:k][~own]))
                    if own.any():
                        lam, coef, rhs, _, ks = ops.scorer.cut_rows(order[:k][own] - lo)
                        assert np.array_equal(out["lam"][:k][own], lam) and np.array_equal(out["rhs"][:k][own], rhs)
                        assert np.array_equal(out["coef"][:k][own], coef[:, :out["coef"].shape[1]])
                        assert np.array_equal(out["ks"][:k][own], ks)
        # every entry visited (SDPCUT_PART_COMBALL): three-field records, merged with obj_improve as secondary key
        for sel in (1, 29, 5000, 8192, 16384):
            for ops in opss:
                ops.scorer.set_point(wl["vars_values"])
            allrec = torch.cat([ops.shard_head(_capi.PART_COMBALL, sel) for ops in opss])
            order, score, _, _ = oracle.rank_arrays(4, O, E, E.shape[0] + 1)      # unreachable quota: every entry visited
            k = min(sel, order.shape[0])
            for r, ops in enumerate(opss):
                ops.shard_finish_enqueue(len(sizes), sel, allrec, sel, fields=3)
            for r, ops in enumerate(opss):
                out = ops.shard_finish_wait()
                g = out["headers"].sum(axis=0)
                assert int(g[0]) == E.shape[0]
                if int(g[4]):
                    # a head of exactly the sort buffers' capacity cannot take a group of equal new scores that straddles
                    # its end whole (this workload is full of duplicates), and equal new scores are not cut by index: the
                    # shard declares its selection void and the caller takes the unfused route
                    assert sel == 8192
                    continue
                assert np.array_equal(out["idx"][:k], order[:k]), (sel, r)
                assert np.array_equal(out["score"][:k], score[:k] + 0.0)
                lo, n = int(bases[r]), sizes[r]
                own = (order[:k] >= lo) & (order[:k] < lo + n)
                assert out["n_own"] == int(own.sum()) and np.array_equal(out["pos"][:out["n_own"]], np.flatnonzero(own))
                if own.any():
                    lam, coef, rhs, _, ks = ops.scorer.cut_rows(order[:k][own] - lo)
                    w = out["n_own"]
                    assert np.array_equal(out["lam"][:w], lam) and np.array_equal(out["rhs"][:w], rhs) and np.array_equal(out["ks"][:w], ks)
    finally:
        for ops in opss:
            ops.scorer.close()


def test_two_ranks_one_gpu_sharded_selection(oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29643", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


_QCQP_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from sdpcutsel_via_nn_amd import _capi, harness, networks
from sdpcutsel_via_nn_amd.distributed import DeviceOps, ShardedSelector, ShardedQCQPRound

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
G = os.path.join(%(root)r, "tests", "golden")
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
g = np.load(os.path.join(G, "inst_qcqp50.npz"))
inst = harness.parse_osil(os.path.join(G, "instances", "q_50_10_25_1.osil"))
n = inst["nb_vars"]
(So, ko), (Sc, kc) = harness.qcqp_covers(inst, 5, _capi.enumerate_cover)


def shard(S, k):
    m = len(k)
    lo, hi = rank * m // world, (rank + 1) * m // world
    sc = _capi.Scorer(0)
    for kk in range(2, 6):
        sc.set_network(kk, *networks.load_network(kk))
    sc.set_instance(n, inst["Q_arr"])
    sc.set_candidates(S[lo:hi], k[lo:hi], global_base=lo)
    return ShardedSelector(DeviceOps(sc, dev), hi - lo)


rnd = ShardedQCQPRound(shard(So, ko), shard(Sc, kc))
assert rnd.sel_obj.n_global == 4 and rnd.sel_cons.n_global == 1377077
for strat in (4, 2, 1):
    q = "s%%d" %% strat
    r = rnd.round(strat, 5000, g["vars"])
    assert r["new_strat"] == int(g[q + "_new_strat"]), (strat, r["new_strat"])
    assert np.array_equal(r["is_obj"], g[q + "_is_obj"]), strat
    ref_score, ref_ids = g[q + "_score"], g[q + "_ids"]
    assert np.abs(r["scores"] - ref_score).max() <= 1e-9 * max(1.0, np.abs(ref_score).max()), strat
    same = r["ids"] == ref_ids
    bad = np.flatnonzero(~same)
    for b in bad:       # the only admissible difference: neighbours whose reference scores agree to 1e-12 (LAPACK noise)
        lo_, hi_ = max(b - 3, 0), min(b + 4, len(ref_ids))
        assert np.ptp(ref_score[lo_:hi_][np.isin(ref_ids[lo_:hi_], r["ids"][lo_:hi_])]) <= 1e-12 * max(1.0, abs(ref_score[b])), (strat, b)
    assert len(bad) <= 10
    assert r["nb_opt_cuts"] == int(g[q + "_nb_opt_cuts"]) and r["nb_cuts_combined"] == int(g[q + "_is_obj"].sum())
    # every rank generated the rows of exactly its own entries, and together they cover the head
    n_obj = int(np.count_nonzero(r["is_obj"])) if strat != 1 else (len(r["rows_obj"]["mine"]) if r["rows_obj"] else 0)
    own = sum(int(p["mine"].sum()) for p in (r["rows_obj"], r["rows_cons"]) if p is not None)
    tot = torch.tensor([own]); dist.all_reduce(tot)
    assert int(tot) == len(r["ids"]) == 5000, (strat, int(tot))
    rc = r["rows_cons"]
    lo_c = rnd.sel_cons.ops.scorer.base
    mine_ids = r["ids"][n_obj:][rc["mine"]]
    assert ((mine_ids >= lo_c) & (mine_ids < lo_c + rnd.sel_cons.n_local)).all()
    lam, coef, rhs, _, ks = rnd.sel_cons.ops.scorer.cut_rows(mine_ids - lo_c)
    assert np.array_equal(rc["lam"], lam) and np.array_equal(rc["coef"], coef[:, :rc["coef"].shape[1]]) and np.array_equal(rc["rhs"], rhs)
    assert (rc["ks"] == 5).all() and r["nb_sdp_cuts"] > 0
for s in (rnd.sel_obj, rnd.sel_cons):
    s.ops.scorer.close()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_ranks_one_gpu_qcqp_round_at_full_size(tmp_path):
    """BASELINE.json configs[4] over shards: q_50_10_25_1, 5-variable sub-problems, both covers split over two
    ranks that share the test box's GPU (gloo); the composed head (cut_select_qcqp.py:79) against the round
    the reference itself ran (tests/golden/inst_qcqp50.npz), rows against sdpcut_cut_rows."""
    script = tmp_path / "qcqp_worker.py"
    script.write_text(_QCQP_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29647", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]

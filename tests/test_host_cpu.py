"""CPU-side tests: the C-ABI library loads and exports what include/sdpcut.h declares, the
product fails loudly without a GPU, the LP harness, the synthetic generator, and the multi-GPU
choreography rehearsed with gloo (world_size 2) on injected device operations."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built_lib():
    from sdpcutsel_via_nn_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built_lib):
    import ctypes
    from sdpcutsel_via_nn_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "sdpcut.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*(sdpcut_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(lib, name), name
    assert _capi.load_library().sdpcut_version() >= 100
    # ... and NOTHING else (-fvisibility=hidden + csrc/exports.map): no C++ internals, no kernel stubs, none of NNs.so's names
    exported = set(ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True).splitlines() if ln.strip())
    assert exported == declared, exported ^ declared


def test_header_constants_match_binding():
    from sdpcutsel_via_nn_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "sdpcut.h")).read()
    vals = dict((k, int(v)) for k, v in re.findall(r"(SDPCUT_\w+)\s*=\s*(-?\d+)", hdr))
    assert (vals["SDPCUT_EIG"], vals["SDPCUT_NN"]) == (_capi.EIG, _capi.NN)
    assert (vals["SDPCUT_STRAT_FEAS"], vals["SDPCUT_STRAT_OPT"], vals["SDPCUT_STRAT_COMB"]) == (1, 2, 4)
    assert vals["SDPCUT_PART_STRONG"] == _capi.PART_STRONG
    assert int(re.search(r"#define SDPCUT_ROW_LD (\d+)", hdr).group(1)) == _capi.ROW_LD
    # (option / statistic codes: from the enum bodies only -- the comments around them quote values too)
    enums = " ".join(re.findall(r"enum\s*\{([^}]*)\}", hdr))
    vals = dict((k, int(v)) for k, v in re.findall(r"(SDPCUT_\w+)\s*=\s*(-?\d+)", enums))
    for name in ("KERNEL", "TIMING", "FUSE_KEYS", "AUTO_REGIME", "FUSED_TAIL", "COOP_LAUNCH"):
        assert vals["SDPCUT_OPT_" + name] == getattr(_capi, "OPT_" + name), name
    for name in ("ROUNDS", "SELECT_FALLBACKS", "SCORED"):
        assert vals["SDPCUT_STAT_" + name] == getattr(_capi, "STAT_" + name), name
    assert (vals["SDPCUT_KERNEL_MFMA"], vals["SDPCUT_KERNEL_SIMPLE"], vals["SDPCUT_KERNEL_VALU"]) == (
        _capi.KERNEL_MFMA, _capi.KERNEL_SIMPLE, _capi.KERNEL_VALU)


def test_product_fails_loudly_without_gpu(built_lib):
    """No CPU fallback: on a box without a HIP device every entry raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sdpcutsel_via_nn_amd as pkg
    with pytest.raises(pkg.SdpCutError, match="no HIP device|no CPU fallback"):
        pkg.Scorer(0)
    cs = pkg.CutSolver()
    cs.set_instance(4, np.zeros(10), [([0, 1, 2], [0, 1, 2, 4, 5, 7], None, None)], dim=3)
    with pytest.raises(pkg.SdpCutError):
        cs._sel_eigcut_by_ordering_on_measure(2, np.zeros(14), 1)
    with pytest.raises(pkg.SdpCutError):
        cs._get_eigendecomp(3, (0.5, 0.5, 0.5), (0.5, 0, 0, 0.5, 0, 0.5), False)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg_dir = os.path.join(ROOT, "sdpcutsel_via_nn_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dirpath, f)


def test_network_packing():
    from sdpcutsel_via_nn_amd import networks
    for k, (hidden, nl) in {2: (64, 4), 3: (50, 4), 4: (50, 4), 5: (64, 5)}.items():
        widths, params = networks.load_network(k)
        d = k * (k + 3) // 2
        assert widths.tolist() == [hidden] * (nl - 1) + [1]
        n = 2 * d + 1 + 3 + hidden * d + hidden + (nl - 2) * (hidden * hidden + hidden) + hidden + 1
        assert params.shape == (n,)
    with pytest.raises(ValueError):
        networks.load_network(6)


def test_synthetic_workload_shape():
    from sdpcutsel_via_nn_amd import synthetic
    wl = synthetic.make_workload(nb_vars=30, k=4, count=5000, seed=3)
    s = wl["set_inds"]
    assert s.shape == (5000, 5) and np.all(s[:, 4] == -1)
    assert np.all(np.diff(s[:, :4], axis=1) > 0) and s[:, :4].min() >= 0 and s[:, :4].max() < 30
    n, L = 30, 465
    vv = wl["vars_values"]
    x, X = vv[L:], vv[:L]
    iu = np.triu_indices(n)
    assert np.all(X <= np.minimum(x[iu[0]], x[iu[1]]) + 1e-15)
    assert np.all(X >= np.maximum(0, x[iu[0]] + x[iu[1]] - 1) - 1e-15)
    wl2 = synthetic.make_workload(nb_vars=30, k=4, count=5000, seed=3)
    assert np.array_equal(wl2["set_inds"], s) and np.array_equal(wl2["vars_values"], vv)


def test_harness_boxqp_and_mccormick(tmp_path):
    """Tiny BoxQP: parser conventions (cut_select_qp.py:313-321), RLT rows (:352-375), LP solve."""
    from sdpcutsel_via_nn_amd import harness
    p = tmp_path / "tiny.in"
    p.write_text("3\n1 -2 3\n2 -4 0\n-4 6 5\n0 5 -8\n")
    inst = harness.parse_boxqp(str(p))
    assert inst["nb_vars"] == 3 and inst["nb_lifted"] == 6
    assert inst["c"].tolist() == [-1, 2, -3]
    assert inst["Q_arr"].tolist() == [-1.0, 4.0, 0.0, -3.0, -5.0, 4.0]
    assert inst["adj"].tolist() == [[True, True, False], [True, True, True], [False, True, True]]
    ptr, ind, val, rhs = harness.mccormick_csr(3, inst["adj"])
    assert len(ptr) - 1 == 2 * 3 + 3 * 2 == len(rhs)
    assert ind[ptr[0]:ptr[1]].tolist() == [0, 6] and val[ptr[0]:ptr[1]].tolist() == [1, -1]         # X00 <= x0
    # row by row against the textbook definition, in the reference's order (cut_select_qp.py:352-375)
    want = []
    for i in range(3):
        Xii = 3 * i - i * (i - 1) // 2
        want += [([Xii, 6 + i], [1, -1], 0), ([Xii, 6 + i], [-1, 2], 1)]
        for j in range(i + 1, 3):
            if inst["adj"][i, j]:
                Xij = Xii + j - i
                want += [([Xij, 6 + i, 6 + j], [-1, 1, 1], 1), ([Xij, 6 + i], [1, -1], 0), ([Xij, 6 + j], [1, -1], 0)]
    got = [(ind[ptr[r]:ptr[r + 1]].tolist(), val[ptr[r]:ptr[r + 1]].tolist(), rhs[r]) for r in range(len(rhs))]
    assert got == want
    lp = harness.boxqp_relaxation(inst)
    lp.solve()
    v = np.asarray(lp.get_values())
    assert v.shape == (9,) and np.all(v >= -1e-9) and np.all(v <= 1 + 1e-9)
    n0 = lp.linear_constraints.get_num()
    lp.linear_constraints.add(lin_expr=[harness.SparsePair([6], [1.0])], rhs=[0.25], senses=["G"])
    assert lp.linear_constraints.get_num() == n0 + 1
    lp.solve()
    assert lp.get_values()[6] >= 0.25 - 1e-9


# ----------------------------------------------------------------------------- multi-GPU choreography
_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from sdpcutsel_via_nn_amd import _capi
from sdpcutsel_via_nn_amd.distributed import ShardedSelector, ShardedQCQPRound, _PAD_ID
from oracle import cutsel_oracle as oracle

class FakeOps(object):
    """Stand-in for the two device calls (ranking of the local shard, merge), backed by the CPU
    oracle, so that the collective choreography can be rehearsed with gloo on a box without GPUs."""
    device = torch.device("cpu")
    class _NoScorer(object):
        def set_point(self, vv):
            pass
    scorer = _NoScorer()
    def __init__(self, obj, lam, base):
        self.obj, self.lam, self.base = obj, lam, base
    def local_head(self, strat, sel_size, count, want_secondary=False):
        obj, lam = self.obj, self.lam
        viol = lam < oracle.THRES_NEG_EIGVAL
        c = dict(nb_violated=0, strong=0, violated=0, nb_positive=0)
        if strat == _capi.PART_STRONG:
            m = (obj > 0) & viol
            key = np.where(m, obj, -np.inf)
            order = np.argsort(-key, kind="stable")[:int(m.sum())]
            score = key[order]
            c["nb_violated"] = int(m.sum())
        else:
            order, score, _, cnt = oracle.rank_arrays(strat, obj, lam, min(sel_size, obj.shape[0]) if strat != 4 else sel_size)
            c.update(cnt)
        total = order.shape[0]
        s = torch.full((count,), float("-inf"), dtype=torch.float64)
        i = torch.full((count,), _PAD_ID, dtype=torch.int64)
        sec = torch.full((count,), float("-inf"), dtype=torch.float64) if want_secondary else None
        w = min(count, total)
        s[:w] = torch.from_numpy(score[:w].copy()); i[:w] = torch.from_numpy(order[:w] + self.base)
        if want_secondary:
            sec[:w] = torch.from_numpy(obj[order[:w]].copy())
        return s, i, sec, total, c
    def merge(self, scores, ids, count_out, secondary=None):
        keys = (ids.numpy(), -secondary.numpy(), -scores.numpy()) if secondary is not None else (ids.numpy(), -scores.numpy())
        o = np.lexsort(keys)[:count_out]
        return scores[o], ids[o]
    # fused round: packed record / merge + rows of the own entries (row = the candidate's lam only)
    max_head = 16384
    def shard_head(self, strat, count, into=None):
        rec = self._shard_head(strat, count)
        if into is not None:
            into.copy_(rec)
            return into
        return rec
    def _shard_head(self, strat, count):
        n = self.obj.shape[0]
        nviol, npos = int((self.lam < oracle.THRES_NEG_EIGVAL).sum()), int((self.obj > 0).sum())
        if strat == _capi.PART_COMBALL:      # PART_COMBALL: every entry visited -> the shard's own combined ranking, obj_improve as third field
            order, score, _, _ = oracle.rank_arrays(4, self.obj, self.lam, n + 1)
            w = min(count, n)
            s = torch.full((count,), float("-inf"), dtype=torch.float64); i = torch.full((count,), _PAD_ID, dtype=torch.int64)
            x = torch.full((count,), float("-inf"), dtype=torch.float64)
            s[:w] = torch.from_numpy(score[:w].copy()); i[:w] = torch.from_numpy(order[:w] + self.base)
            x[:w] = torch.from_numpy(self.obj[order[:w]].copy())
            hdr = [n, nviol, npos, w, 0, 0, 0, 0]
            return torch.cat([torch.tensor(hdr, dtype=torch.int64), s.view(torch.int64), i, x.view(torch.int64)])
        s, i, _, total, _ = self.local_head(strat, count, count)
        hdr = [n if strat == 2 else total, nviol, npos, min(count, total), 0, 0, 0, 0]
        return torch.cat([torch.tensor(hdr, dtype=torch.int64), s.view(torch.int64), i])
    def rows_of(self, ids):
        n = self.obj.shape[0]
        mine = (ids >= self.base) & (ids < self.base + n)
        c = int(mine.sum())
        return mine, self.lam[ids[mine] - self.base], np.zeros((c, 9)), np.zeros(c), np.full(c, 3, dtype=np.int32)
    def shard_finish_enqueue(self, world, count, allrec, sel, fields=2, pitch_words=0, offset_words=0):
        a = allrec.view(world, -1).numpy()
        if pitch_words:
            assert a.shape[1] == pitch_words
            a = a[:, offset_words:offset_words + 8 + fields * count]
        s = np.ascontiguousarray(a[:, 8:8 + count]).reshape(-1).view(np.float64)
        i = np.ascontiguousarray(a[:, 8 + count:8 + 2 * count]).reshape(-1)
        if fields == 3:
            x = np.ascontiguousarray(a[:, 8 + 2 * count:]).reshape(-1).view(np.float64)
            o = np.lexsort((i, -x, -s))[:sel]
        else:
            o = np.lexsort((i, -s))[:sel]
        mine, lam_m, _, _, _ = self.rows_of(i[o])
        w = int(mine.sum())
        self._pending = dict(headers=a[:, :8].copy(), idx=i[o], score=s[o].copy(), lam=lam_m, coef=np.zeros((w, 9)), rhs=np.zeros(w),
                             ks=np.full(w, 3, dtype=np.int32), pos=np.flatnonzero(mine).astype(np.int32), n_own=w)
    def shard_finish_wait(self):
        return self._pending

def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    rng = np.random.default_rng(42)
    for case, (n_each, pos_frac, sel) in enumerate([(3000, 0.5, 400), (3000, 0.01, 400), (500, 0.0, 1200), (64, 0.9, 5)]):
        N = n_each * world
        obj = np.round(rng.normal(size=N), 1)                  # many exact ties
        obj = np.where(rng.uniform(size=N) < pos_frac, np.abs(obj) + 0.1, -np.abs(obj))
        lam = np.round(rng.normal(size=N) * 0.3, 2) - 0.001
        lo = rank * n_each
        ops = FakeOps(obj[lo:lo + n_each], lam[lo:lo + n_each], lo)
        sel_obj = ShardedSelector(ops, n_each)
        assert sel_obj.n_global == N
        for strat in (1, 2, 4):
            r = sel_obj.select(strat, sel)
            order, score, new_strat, cnt = oracle.rank_arrays(strat, obj, lam, sel)
            k = min(sel, order.shape[0])
            assert np.array_equal(r["ids"].numpy(), order[:k]), (case, strat, rank)
            assert np.array_equal(r["scores"].numpy(), score[:k]), (case, strat)
            assert r["new_strat"] == new_strat, (case, strat)
            if strat == 4:
                assert r["counters"]["strong"] == cnt["strong"] and r["counters"]["violated"] == cnt["violated"]
            # the fused round: same head, plus the rows of this rank's own entries
            q = sel_obj.select_round(strat, sel)
            assert np.array_equal(q["ids"], order[:k]), (case, strat, rank)
            assert np.array_equal(q["scores"], score[:k]), (case, strat)
            assert q["new_strat"] == new_strat, (case, strat)
            own = (order[:k] >= lo) & (order[:k] < lo + n_each)
            assert np.array_equal(q["mine"], own), (case, strat)
            assert np.array_equal(q["lam"], lam[order[:k][own]]) and q["ks"].shape[0] == int(own.sum())
            if strat == 4:
                assert q["counters"]["strong"] == cnt["strong"] and q["counters"]["violated"] == cnt["violated"]
    # QCQP composition over shards (cut_select_qcqp.py:79): objective cover ranked with the strategy, constraints-only
    # cover with feasibility, head = (A + B)[0:sel]
    for case, (na, nb, sel) in enumerate([(3, 900, 100), (40, 300, 25), (0, 500, 60), (16, 0, 10)]):
        Na, Nb = na * world, nb * world
        oa, la = np.abs(np.round(rng.normal(size=Na), 1)) + 0.1, np.round(rng.normal(size=Na) * 0.3, 2) - 0.001
        ob, lb = np.round(rng.normal(size=Nb), 1), np.round(rng.normal(size=Nb) * 0.3, 2) - 0.001
        sa = ShardedSelector(FakeOps(oa[rank * na:(rank + 1) * na], la[rank * na:(rank + 1) * na], rank * na), na)
        sb = ShardedSelector(FakeOps(ob[rank * nb:(rank + 1) * nb], lb[rank * nb:(rank + 1) * nb], rank * nb), nb)
        rnd = ShardedQCQPRound(sa, sb)
        for strat in (1, 2, 4):
            r = rnd.round(strat, sel, None)
            order_a, score_a, new_strat, _ = oracle.rank_arrays(strat, oa, la, sel) if Na else (np.empty(0, np.int64), np.empty(0), strat, None)
            n_obj = min(sel, order_a.shape[0])
            order_b, score_b, _, _ = oracle.rank_arrays(1, ob, lb, sel) if Nb else (np.empty(0, np.int64), np.empty(0), 1, None)
            n_b = min(sel - n_obj, order_b.shape[0])
            assert np.array_equal(r["ids"], np.concatenate([order_a[:n_obj], order_b[:n_b]])), (case, strat, rank)
            assert np.array_equal(r["scores"], np.concatenate([score_a[:n_obj], score_b[:n_b]]) + 0.0), (case, strat)
            assert r["new_strat"] == new_strat and r["nb_cuts_combined"] == (n_obj if strat != 1 else 0)
            assert r["nb_opt_cuts"] == (int((score_a[:n_obj] > 1000.0).sum()) if strat != 1 else 0)
            lam_sel = np.concatenate([la[order_a[:n_obj]], lb[order_b[:n_b]]])
            assert r["nb_sdp_cuts"] == int((lam_sel < -1e-15).sum()), (case, strat)
    dist.destroy_process_group()
    print("rank", rank, "ok")

main()
'''


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_selection_gloo(oracle, tmp_path, world):
    """world_size-2 and -8 rehearsal (gloo): per-shard heads + all-gather + merge reproduce the
    single-list ranking head bit-exactly for strategies 1, 2 and 4 (both regimes of the
    combined scan, heavy ties included).  Eight ranks = the shape of the driver's largest run."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29641 + world), OMP_NUM_THREADS="1")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o

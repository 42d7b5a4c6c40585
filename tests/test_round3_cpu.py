"""CPU-side tests of the round-3 host logic (no GPU): the rank list that carries a fused round's assembled cuts, its
heads and concatenations, the hand-over to the LP's row store, the build stamp without a compiler, the C-ABI's new
symbols and struct."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT


class _FakeScorer(object):
    base = 0
    round_count = 5

    def rank(self, *a, **k):
        raise AssertionError("the device must not be asked for anything in these tests")


def _rank_list(n_head=10, n_total=50, strat=4):
    """a RankList over a fake binding whose fused round produced cuts for head entries 0, 2, 3, 5, 6, 9 (2, 3 or 5 non-zeros)"""
    from sdpcutsel_via_nn_amd.cut_solver import RankList, _Binding
    S = np.array([[i % 7, i % 7 + 1, i % 7 + 2, -1, -1] for i in range(n_total)], dtype=np.int32)
    ks = np.full(n_total, 3, dtype=np.int32)
    b = _Binding(_FakeScorer(), None, 12, 78)
    b.set_arr, b.ks, b.n_at_bind = S, ks, n_total
    b.point_token = 1
    vv = np.linspace(0, 1, 90)
    idx = np.arange(100, 100 + n_head, dtype=np.int64) % n_total
    score = np.array([1009.0, 1008.0, 1007.0, 1006.0, 1005.0, 0.5, 0.0, -1.0, -2.0, -3.0])[:n_head]
    row_entry = np.array([0, 2, 3, 5, 6, 9], dtype=np.int32)
    lens = np.array([2, 3, 5, 2, 3, 5])
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    fused = dict(idx=idx, score=score, lam=-np.ones(n_head), ks=ks[idx], set_inds=S[idx], row_entry=row_entry, indptr=indptr,
                 indices=np.arange(indptr[-1], dtype=np.int32), values=np.arange(indptr[-1], dtype=np.float64) / 10, rhs=-np.arange(6.0))
    return RankList(None, b, 1 if strat == 1 else 2, n_total, vv, idx.copy(), score.copy(), strat=strat, sel_size=n_head, fused=fused), b, vv


def test_fused_rows_are_a_prefix_of_the_block():
    rl, b, vv = _rank_list()
    ptr, ind, val, rhs = rl.fused_rows(10)
    assert ptr.tolist() == [0, 2, 5, 10, 12, 15, 20] and rhs.tolist() == [-0.0, -1.0, -2.0, -3.0, -4.0, -5.0] and len(ind) == len(val) == 20
    # the cuts of the first m entries: rows whose head position is < m
    for m, rows in ((1, 1), (2, 1), (3, 2), (4, 3), (6, 4), (9, 5)):
        ptr, ind, val, rhs = rl.fused_rows(m)
        assert rhs.shape[0] == rows and ptr.shape[0] == rows + 1 and len(ind) == ptr[-1]
    # strong_only (cut_select_qp.py:725-726): stop at the first score <= 0 -> entries 0..5 -> 4 cuts
    assert rl.fused_rows(10, strong_only=True)[3].shape[0] == 4
    # another LP point than the one the round ranked at: no fused rows (the same object or equal values are fine)
    assert rl.fused_rows(10, vars_values=vv) is not None and rl.fused_rows(10, vars_values=vv.copy()) is not None
    assert rl.fused_rows(10, vars_values=vv + 1e-9) is None
    # longer than the head, or the scorer has run another round since: no fused rows
    assert rl.fused_rows(11) is None
    b.scorer.round_count += 1
    assert rl.fused_rows(10) is None


def test_heads_and_concatenations_keep_the_link_to_the_block():
    from sdpcutsel_via_nn_amd.cut_solver import RankListHead, _Concat
    rl, b, vv = _rank_list()
    head = rl[0:6]
    assert isinstance(head, RankListHead) and len(head) == 6 and head.parent is rl
    assert isinstance(head[0:3], RankListHead) and len(head[0:3]) == 3
    ent = list(head)
    assert [e[0] for e in ent] == rl.ids(6).tolist() and all(isinstance(e[0], int) and isinstance(e[1], float) for e in ent)
    assert ent[0][2] == tuple(vv[78 + i] for i in (0, 1, 2))          # curr_pt of candidate 100 % 50 = 0, index set (0, 1, 2)
    assert head == ent and head[2] == ent[2] and head[1:3] == ent[1:3]
    # (A + B)[0:sel_size], cut_select_qcqp.py:79, on lists whose heads cover the slice (a slice beyond a head asks the device
    # for the full ranking -- _FakeScorer.rank raises)
    first, _, _ = _rank_list(n_total=10)
    other, _, _ = _rank_list(n_total=10, strat=1)
    cat = first + other
    assert isinstance(cat, _Concat) and len(cat) == 20
    sl = cat[0:13]
    assert isinstance(sl, _Concat) and len(sl) == 13 and [len(p) for p in sl._parts] == [10, 3]
    assert isinstance(sl._parts[1], RankListHead) and sl._parts[1].parent is other
    assert isinstance(cat[0:4]._parts[0], RankListHead)
    assert len(list(cat[8:12])) == 4
    with pytest.raises(AssertionError, match="device"):
        rl[0:11]


def test_assembled_cuts_reach_both_kinds_of_row_store():
    from sdpcutsel_via_nn_amd import harness
    from sdpcutsel_via_nn_amd.cut_solver import GpuCutSelectionMixin
    rl, b, vv = _rank_list()

    class Solver(GpuCutSelectionMixin):
        pass
    s = Solver()
    s._my_prob = harness.LinearRelaxation(np.zeros(90))
    nb = s._gen_eigcuts_selected(4, 6, rl, vars_values=vv)
    store = s._my_prob.linear_constraints
    assert nb == 4 == store.get_num() and store.senses == ["G"] * 4 and store.rhs == [-0.0, -1.0, -2.0, -3.0]
    assert [r.ind for r in store.rows] == [[0, 1], [2, 3, 4], [5, 6, 7, 8, 9], [10, 11]]
    # the store owns copies: the pinned block is reused by the next round
    rl._fused["values"][:] = -7.0
    assert store.rows[0].val == [0.0, 0.1]

    class RefStore(object):                          # the reference's LP surface (cplex): add(lin_expr=, rhs=, senses=) only
        def __init__(self):
            self.calls = []

        def add(self, lin_expr=(), rhs=(), senses=()):
            self.calls.append((list(lin_expr), list(rhs), list(senses)))

    class Prob(object):
        linear_constraints = RefStore()
    rl2, _, vv2 = _rank_list()
    s2 = Solver()
    s2._sparse_pair = harness.SparsePair
    s2._my_prob = Prob()
    assert s2._gen_eigcuts_selected(4, 10, rl2[0:10], strong_only=True, vars_values=vv2) == 4
    (rows, rhs, senses), = Prob.linear_constraints.calls
    assert [r.ind for r in rows] == [[0, 1], [2, 3, 4], [5, 6, 7, 8, 9], [10, 11]] and rhs == [-0.0, -1.0, -2.0, -3.0] and senses == ["G"] * 4
    assert all(isinstance(v, float) for r in rows for v in r.val) and all(isinstance(i, int) for r in rows for i in r.ind)


def test_build_without_a_compiler_uses_a_matching_shipped_library(monkeypatch, tmp_path):
    from sdpcutsel_via_nn_amd import build
    build.build(verbose=False)
    monkeypatch.setattr(build, "HIPCC", str(tmp_path / "no-such-hipcc"))
    assert build._compiler_id().startswith("unavailable")
    assert build.build(verbose=False) == build.LIB           # sources match the stamp: the shipped library is the answer
    src = os.path.join(build.CSRC, "shard.hip")
    text = open(src).read()
    try:
        with open(src, "a") as f:
            f.write("\n// changed\n")
        with pytest.raises(RuntimeError, match="no hipcc"):
            build.build(verbose=False)
    finally:
        with open(src, "w") as f:
            f.write(text)
    monkeypatch.undo()
    build.build(verbose=False)                                # (nothing is stale: no compiler run either)


def test_round_csr_struct_matches_the_header():
    """sdpcut_round_csr_t as ctypes sees it == as a C compiler sees the header"""
    import subprocess
    from sdpcutsel_via_nn_amd import _capi
    src = tmp = os.path.join(ROOT, "tests", "_csr_layout.c")
    exe = os.path.join(ROOT, "tests", "_csr_layout")
    fields = [f[0] for f in _capi.RoundCsr._fields_]
    with open(src, "w") as f:
        f.write('#include <stdio.h>\n#include <stddef.h>\n#include "../include/sdpcut.h"\nint main(void) { printf("%zu", sizeof(sdpcut_round_csr_t));\n'
                + "".join('printf(" %%zu", offsetof(sdpcut_round_csr_t, %s));\n' % n for n in fields) + "return 0; }\n")
    try:
        subprocess.check_call(["gcc", "-o", exe, src])
        out = [int(v) for v in subprocess.check_output([exe]).split()]
    finally:
        for p in (src, exe):
            if os.path.exists(p):
                os.remove(p)
    assert out[0] == ctypes.sizeof(_capi.RoundCsr)
    assert out[1:] == [getattr(_capi.RoundCsr, n).offset for n in fields]


# ----------------------------------------------------------------------------- the QCQP pairing, without a device
class _RecordingScorer(object):
    """stands in for _capi.Scorer: records the calls of the two-halves API and answers with a canned round"""
    log = []
    base = 0

    def __init__(self, name, n):
        self.name, self.n, self.round_count, self.pending = name, n, 0, None

    def set_option(self, opt, value):
        assert self.pending is None
        _RecordingScorer.log.append((self.name, "option", opt, value))

    def round_csr_begin(self, strat, sel, point=None):
        assert self.pending is None, "a second begin on a pending scorer"
        self.round_count += 1
        self.pending = object()
        self._args = (strat, sel, np.array(point))
        _RecordingScorer.log.append((self.name, "begin", strat, sel))
        return self.pending

    def round_csr_end(self, copy=False):
        assert self.pending is not None
        self.pending = None
        strat, sel, _ = self._args
        _RecordingScorer.log.append((self.name, "end", strat, sel))
        w = min(sel, self.n)
        z = np.zeros
        return dict(idx=np.arange(w, dtype=np.int64), score=np.linspace(2000.0, 1001.0, w), lam=-np.ones(w), ks=np.full(w, 3, np.int32),
                    set_inds=z((w, 5), np.int32), row_entry=np.arange(w, dtype=np.int32), indptr=np.arange(w + 1, dtype=np.int32) * 2,
                    indices=z(2 * w, np.int32), values=z(2 * w), rhs=z(w), n_total=self.n, new_strat=strat,
                    counters=dict(nb_violated=w, strong=w, violated=w, nb_positive=w))

    def get_candidates(self, idx):
        raise AssertionError("not needed")


def test_the_second_list_of_a_qcqp_round_is_begun_with_the_first(monkeypatch):
    """The pairing of cut_select_qcqp.py:64-76 as the mixin learns it, call by call, with recording scorers: round 1 runs the two
    lists one after the other and notes the pair; from round 2 on the LONGER list (the follower here) is begun first, the short
    leader's stream is prioritised once, and the follower's own call only ends its round; a follower asked about another point
    drops the speculative round and the pair is unlearnt."""
    from sdpcutsel_via_nn_amd import _capi
    from sdpcutsel_via_nn_amd.cut_solver import CutSolver, _Binding
    cs = CutSolver()
    cs._nb_vars, cs._nb_lifted, cs._dim = 12, 78, 3
    agg_a, agg_b = [([0, 1, 2],)] * 40, [([0, 1, 2],)] * 4000
    sa, sb = _RecordingScorer("A", 40), _RecordingScorer("B", 4000)
    ba, bb = _Binding(sa, agg_a, 12, 78), _Binding(sb, agg_b, 12, 78)
    for b, agg in ((ba, agg_a), (bb, agg_b)):
        b.n_at_bind, b.serial = len(agg), 0
    cs._gpu_bindings = {id(agg_a): ba, id(agg_b): bb}
    log = _RecordingScorer.log
    del log[:]
    rng = np.random.default_rng(0)

    def round_(vv, second_point=None):
        cs._agg_list = agg_a
        cs._sel_eigcut_by_ordering_on_measure(2, vv, 1)
        cs._agg_list = agg_b
        cs._sel_eigcut_by_ordering_on_measure(1, vv if second_point is None else second_point, 1)

    p1, p2, p3, p4 = (rng.random(90) for _ in range(4))
    round_(p1)
    assert [(e[0], e[1]) for e in log] == [("A", "begin"), ("A", "end"), ("B", "begin"), ("B", "end")]
    assert ba.follower is not None and ba.follower[0] is bb and bb.leader is ba
    del log[:]
    round_(p2)
    assert log[0] == ("A", "option", _capi.OPT_STREAM_PRIORITY, 1)
    assert [(e[0], e[1]) for e in log[1:]] == [("B", "begin"), ("A", "begin"), ("A", "end"), ("B", "end")]
    del log[:]
    round_(p3)                                            # (the priority is set once)
    assert [(e[0], e[1]) for e in log] == [("B", "begin"), ("A", "begin"), ("A", "end"), ("B", "end")]
    del log[:]
    round_(p4, second_point=p1)                           # the pattern breaks: B is asked about another point
    assert [(e[0], e[1]) for e in log] == [("B", "begin"), ("A", "begin"), ("A", "end"), ("B", "end"), ("B", "begin"), ("B", "end")]
    assert bb.wasted == 1 and ba.follower is None and bb.leader is None and sa.pending is None and sb.pending is None
    # switched off: never paired
    cs2 = CutSolver()
    cs2._gpu_overlap = False
    cs2._nb_vars, cs2._nb_lifted, cs2._dim = 12, 78, 3
    ba2, bb2 = _Binding(_RecordingScorer("A", 40), agg_a, 12, 78), _Binding(_RecordingScorer("B", 4000), agg_b, 12, 78)
    for b, agg in ((ba2, agg_a), (bb2, agg_b)):
        b.n_at_bind, b.serial = len(agg), 0
    cs2._gpu_bindings = {id(agg_a): ba2, id(agg_b): bb2}
    for vv in (p1, p2):
        cs2._agg_list = agg_a
        cs2._sel_eigcut_by_ordering_on_measure(2, vv, 1)
        cs2._agg_list = agg_b
        cs2._sel_eigcut_by_ordering_on_measure(1, vv, 1)
    assert ba2.follower is None and bb2.leader is None


def test_count_above_from_the_head():
    """RankList.count_above without a device: answered from the head when the list's order (and, under the combined strategy,
    the marking rule of cut_select_qp.py:611) makes that exact; the full ranking is asked for otherwise (the fake raises)."""
    rl, b, vv = _rank_list()                              # head scores 1009 .. 1005, 0.5, 0, -1, -2, -3 of a list of 50
    assert rl.count_above(1000.0) == 5 and rl.count_above(0.25) == 6 and rl.count_above(-3.0) == 9
    with pytest.raises(AssertionError):
        rl.count_above(-3.5)                              # entries behind the head may exceed it: needs the complete ranking
    rl2, _, _ = _rank_list(n_head=5)                      # a head of marked entries only (combined strategy)
    assert rl2.count_above(1000.0) == 5                   # the entries behind it have obj_improve <= 1005 - 1000
    with pytest.raises(AssertionError):
        rl2.count_above(4.0)                              # ... which may well exceed 4

"""CPU-side tests of the round-2 additions: the counter-based candidate generator (numpy twin vs
the published Philox known answers vs the C header), the reference's own FFI symbols in the product
library, composition of the GPU mixin with the REAL reference classes (when /root/reference is
present), lazy rank-list concatenation, the round driver and the content-keyed build."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference"


# ----------------------------------------------------------------------------- Philox generator
def test_philox_known_answers():
    """Random123 known-answer vectors of philox4x32-10 (kat_vectors: counter, key -> output)."""
    from sdpcutsel_via_nn_amd import synthetic as s
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = s.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_philox_index_sets_twin_equals_c_header(tmp_path):
    """csrc/philox.h (the arithmetic the device kernel runs) compiled for the host == numpy twin."""
    from sdpcutsel_via_nn_amd import synthetic as s
    src = tmp_path / "p.cpp"
    src.write_text('#include "%s"\nextern "C" void sets(unsigned long long seed, const unsigned long long *ids, long n_ids, '
                   'int nv, int k, int *out) { for (long i = 0; i < n_ids; ++i) philox_index_set(seed, ids[i], nv, k, out + 5 * i); }\n'
                   % os.path.join(ROOT, "sdpcutsel_via_nn_amd", "csrc", "philox.h"))
    so = tmp_path / "p.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    rng = np.random.default_rng(0)
    ids = np.concatenate([np.arange(20000), rng.integers(0, 2 ** 40, 20000)]).astype(np.uint64)
    for nv, k, seed in ((1000, 3, 7), (100, 2, 7), (30, 4, 123456789012345), (12, 5, 9), (10, 5, 1)):
        out = np.empty((ids.shape[0], 5), dtype=np.int32)
        lib.sets(ctypes.c_ulonglong(seed), ids.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), ctypes.c_long(ids.shape[0]),
                 nv, k, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        twin = s.philox_index_sets(nv, k, ids, seed)
        assert np.array_equal(out, twin), (nv, k)
        assert np.all(np.diff(twin[:, :k], axis=1) > 0) and twin[:, :k].min() >= 0 and twin[:, :k].max() < nv
        assert np.all(twin[:, k:] == -1)
    # uniform over the k-subsets (n = 10, k = 5: 252 subsets, attempts are redrawn often): chi-square
    t = s.philox_index_sets(10, 5, np.arange(252 * 400), 1)
    code = (t[:, :5] * (10 ** np.arange(5))).sum(axis=1)
    _, counts = np.unique(code, return_counts=True)
    assert counts.shape[0] == 252
    chi2 = ((counts - 400.0) ** 2 / 400.0).sum()
    assert chi2 < 251 + 5 * np.sqrt(2 * 251), chi2


# ----------------------------------------------------------------------------- boundary
def test_product_library_exports_the_reference_ffi():
    """SURVEY 8 b: the six symbols of NNs.so (cut_select_qp.py:297-303) come from the PRODUCT -- since r5 from its NNs.so
    replacement libsdpcut_nns.so (include/sdpcut_nns.h), which exports nothing else and binds to the GPU library privately."""
    import subprocess
    from sdpcutsel_via_nn_amd import _capi, build
    build.build(verbose=False)
    lib = ctypes.CDLL(_capi.NNS_LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "sdpcut_nns.h")).read()
    for name in _capi.COMPAT_SYMBOLS:
        assert hasattr(lib, name), name
        assert name + "(" in hdr
    exported = set(ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", _capi.NNS_LIB_PATH], text=True).splitlines() if ln.strip())
    assert exported == set(_capi.COMPAT_SYMBOLS), exported
    needed = subprocess.check_output(["readelf", "-d", _capi.NNS_LIB_PATH], text=True)
    assert "amdhip" not in needed and "sdpcut_hip" not in needed      # no HIP runtime of its own, the GPU library is dlopen'ed RTLD_LOCAL
    # without a GPU the call says so and returns NaN (the signature has no error channel)
    import torch
    if not torch.cuda.is_available():
        f = lib.neural_net_3D
        f.restype = ctypes.c_double
        buf = (ctypes.c_double * 9)(*([0.5] * 9))
        assert np.isnan(f(buf))


def test_build_is_keyed_on_content(tmp_path):
    """Touching a source (new mtime, same bytes) does not trigger a rebuild; the stamp lists every unit."""
    import json
    from sdpcutsel_via_nn_amd import build
    build.build(verbose=False)
    stamp = json.load(open(build.STAMP))
    assert set(build.SOURCES) <= set(stamp) and "__lib__" in stamp
    src = os.path.join(build.CSRC, "shard.hip")
    obj = os.path.join(build.CSRC, "shard.o")
    before = os.path.getmtime(obj)
    os.utime(src, None)
    build.build(verbose=False)
    assert os.path.getmtime(obj) == before


def _reference_modules():
    if not os.path.isdir(REF):
        pytest.skip("reference not present (GPU box)")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    cwd = os.getcwd()
    try:
        return make_golden.import_reference()
    finally:
        os.chdir(cwd)


def test_mixin_drops_into_the_reference_classes():
    """`make_dropin_classes` on the real reference modules (solver packages replaced by the inert
    stand-ins of tests/golden/make_golden.py): every hot-path call site of BOTH loops resolves to the
    mixin -- including the `super()` calls inside CutSolverQCQP (cut_select_qcqp.py:41, :66-76) and
    the name-mangled triangle privates (cut_select_qp.py:141, :185)."""
    import sdpcutsel_via_nn_amd as pkg
    qp, qcqp = _reference_modules()
    Mixin = pkg.GpuCutSelectionMixin
    G, GQ = pkg.make_dropin_classes(qp, qcqp)
    hot = ["_load_neural_nets", "_sel_eigcut_by_ordering_on_measure", "_gen_eigcuts_selected", "_get_eigendecomp"]
    assert [c.__name__ for c in G.__mro__] == ["GpuCutSolver", "GpuCutSelectionMixin", "CutSolver", "object"]
    assert [c.__name__ for c in GQ.__mro__] == ["GpuCutSolverQCQP", "CutSolverQCQP", "GpuCutSolver", "GpuCutSelectionMixin",
                                                "CutSolver", "object"]
    g, gq = G(), GQ()
    for name in hot:
        assert getattr(g, name).__func__ is getattr(Mixin, name)
        assert getattr(gq, name).__func__ is getattr(Mixin, name)
        # what `super().<name>` means inside the reference's QCQP loop
        assert getattr(super(qcqp.CutSolverQCQP, gq), name).__func__ is getattr(Mixin, name)
    for mangled, ours in (("_CutSolver__preprocess_triangle_ineq", "_preprocess_triangle_ineq"),
                          ("_CutSolver__separate_and_add_triangle", "_separate_and_add_triangle")):
        assert getattr(g, mangled).__func__ is getattr(Mixin, ours)
        assert getattr(gq, mangled).__func__ is getattr(Mixin, ours)
    # everything else stays the reference's
    assert g.cut_select_algo.__func__ is qp.CutSolver.cut_select_algo
    assert gq.cut_select_algo.__func__ is qcqp.CutSolverQCQP.cut_select_algo
    assert super(qcqp.CutSolverQCQP, gq)._add_mccormick_to_instance.__func__ is qp.CutSolver._add_mccormick_to_instance
    assert gq._BIG_M == 1000 and g._THRES_NEG_EIGVAL == -1e-15
    # the naive order (mixin in front of CutSolverQCQP) would leave the QCQP loop's super() calls on the CPU
    Naive = type("Naive", (Mixin, qcqp.CutSolverQCQP), {})
    assert getattr(super(qcqp.CutSolverQCQP, Naive()), hot[1]).__func__ is getattr(qp.CutSolver, hot[1])
    # and the composed object runs the reference's own setup code up to the first GPU call
    import torch
    if not torch.cuda.is_available():
        g._dim = 3
        g._load_neural_nets()                       # weights only, no device yet
        cwd = os.getcwd()
        os.chdir(REF)
        try:
            g._CutSolver__parse_boxqp_into_cplex("spar020-100-1")
            assert g._get_sdp_vertex_cover(3) == 1051
        finally:
            os.chdir(cwd)
        with pytest.raises(pkg.SdpCutError):
            g._sel_eigcut_by_ordering_on_measure(2, np.full(230, 0.5), 1)


# ----------------------------------------------------------------------------- host containers
def test_lazy_concat_and_agg_shuffle():
    from sdpcutsel_via_nn_amd.cut_solver import AggArrays, _Concat
    a, b = list(range(10)), [10, 11, 12]
    c = _Concat([a, b])
    assert len(c) == 13 and c[0:4] == [0, 1, 2, 3] and c[8:12] == [8, 9, 10, 11] and c[10:99] == [10, 11, 12]
    assert c[11] == 11 and c[-1] == 12 and list(c) == a + b and (c + [13])[12:] == [12, 13]
    assert c[0:0] == [] and c[::5] == [0, 5, 10]
    with pytest.raises(IndexError):
        c[13]
    # AggArrays.shuffle == np.random.shuffle of the equivalent list (same generator state -> same order)
    rng = np.random.default_rng(1)
    S = np.sort(rng.integers(0, 30, (200, 5)).astype(np.int32), axis=1)
    ks = rng.integers(2, 6, 200).astype(np.int32)
    agg = AggArrays(S.copy(), ks.copy(), 30)
    as_list = [agg[i] for i in range(200)]
    np.random.seed(7)
    np.random.shuffle(as_list)
    np.random.seed(7)
    agg.shuffle()
    assert agg.serial == 1 and [agg[i][0] for i in range(200)] == [e[0] for e in as_list]


def test_round_driver_bookkeeping():
    from sdpcutsel_via_nn_amd import harness

    class FakeLP(object):
        def __init__(self):
            self.v, self.rows = 10.0, 0
        def solve(self):
            self.v = 10.0 - 4.0 * (1 - 0.5 ** self.rows)
        def get_values(self):
            return [self.v]
        def get_objective_value(self):
            return self.v

    lp = FakeLP()
    seen = []

    def separate(r, point):
        seen.append((r, float(point[0])))
        lp.rows += 1
        return {"sdp": 3 * r}

    ticks = iter(range(1000))
    log = harness.run_cut_rounds(lp, separate, 5, setup_s=100.0, clock=lambda: next(ticks))
    assert log.bounds == [10.0, 8.0, 7.0, 6.5, 6.25, 6.125] and [r for r, _ in seen] == [1, 2, 3, 4, 5]
    assert seen[0][1] == 10.0 and seen[1][1] == 8.0             # each round separates the previous solve's point
    assert log.column("sdp") == [3, 6, 9, 12, 15] and log.column("tri") == [0] * 5
    assert log.solve_s[0] == 101.0 and log.solve_s[1:] == [1.0] * 5 and log.separation_s == [1.0] * 5
    lp2 = FakeLP()
    log2 = harness.run_cut_rounds(lp2, lambda r, p: (setattr(lp2, "rows", lp2.rows + 1) or {}), 20, stop_tol=0.2)
    # improvement / gap closed so far: 1/3, 1/7 < 0.2 -> stops before round 4
    assert len(log2.bounds) == 4

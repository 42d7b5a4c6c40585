"""Round-5 CPU tests: the bench line's parity object, the CPU baseline's core count, the exact-eigenvalue helper of the
trajectory replay, the launcher's refusal under a profiler preload."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_selection_parity_object(oracle):
    """bench.selection_parity (what BENCH_rNN.json's `parity` comes from): identical rounds say so; a swapped pair, a foreign id and a
    score deviation are counted."""
    import bench
    rng = np.random.default_rng(4)
    n, sel = 20000, 500
    obj = rng.normal(size=n) * 10
    lam = rng.normal(size=n) * 0.1
    max_elem = np.full(n, 150.0)
    for strat in (1, 2, 4):
        order, scores, new_strat, _ = oracle.rank_arrays(strat, None if strat == 1 else obj, None if strat == 2 else lam, sel)
        rec = dict(idx=order[:sel].copy(), score=scores[:sel].copy(), new_strat=new_strat,
                   eig=None if strat == 2 else lam.copy(), obj=None if strat == 1 else obj.copy())
        par = bench.selection_parity(oracle, strat, None if strat == 1 else obj, None if strat == 2 else lam, max_elem, sel, rec)
        assert par["topk_identical"] and par["positions_differing"] == 0 and par["ids_one_side_only"] == 0 and par["new_strategy_identical"]
        assert par["head"] == par["gpu_head"] == sel and par["max_abs_d_head_score"] == 0.0
        bad = dict(rec, idx=rec["idx"].copy())
        bad["idx"][[3, 4]] = bad["idx"][[4, 3]]                     # two neighbours swapped
        bad["idx"][-1] = order[sel + 7]                              # an id the oracle's head does not hold
        if strat != 2:
            bad["eig"] = lam + 3e-15
        par = bench.selection_parity(oracle, strat, None if strat == 1 else obj, None if strat == 2 else lam, max_elem, sel, bad)
        assert not par["topk_identical"] and par["positions_differing"] == 3 and par["ids_one_side_only"] == 2
        if strat != 2:
            assert abs(par["max_abs_d_eig"] - 3e-15) < 1e-16


def test_host_cores_is_what_the_process_may_use():
    import bench
    cores, how = bench.host_cores()
    assert 1 <= cores <= len(os.sched_getaffinity(0)) and "sched_getaffinity" in how


def test_exact_lambda_min_helper():
    """tests/exact_eig.py: integer characteristic polynomial + 80-digit root, against LAPACK on generic and on exactly singular
    lifted matrices (a McCormick-vertex matrix has lambda_min = 0 exactly; LAPACK returns 1e-17-level noise)."""
    import exact_eig
    rng = np.random.default_rng(2)
    for k in (2, 3, 4, 5):
        for _ in range(25):
            x, X = rng.random(k), rng.random(k * (k + 1) // 2) * 0.5
            A = exact_eig.lifted_full(k, x, X)
            lap = np.linalg.eigvalsh(A)[0]
            t = exact_eig.exact_lambda_min(A, lap)
            assert abs(float(t) - lap) <= 4e-15
            cf = exact_eig.charpoly_exact(A)
            # the polynomial vanishes at the exact root to 60 digits, and LAPACK's other eigenvalues are near-roots too
            from decimal import Decimal
            p = Decimal(0)
            for a in cf:
                p = p * t + Decimal(a.numerator) / Decimal(a.denominator)
            assert abs(p) < Decimal(10) ** -50
    A = exact_eig.lifted_full(3, [0.5, 0.5, 0.5], [0.5, 0.0, 0.5, 0.5, 0.0, 0.5])
    assert abs(exact_eig.exact_lambda_min(A, np.linalg.eigvalsh(A)[0])) < 1e-50


def test_self_launch_refuses_under_a_profiler_preload():
    """`python bench.py --gpus N` must not start ranks from a process a profiler has attached to (the preload
    initialises the GPU in the parent; the exec that follows is what this pool's machines refuse)."""
    env = dict(os.environ, ROCP_TOOL_LIBRARIES="librocprofiler-sdk-tool.so", SDPCUT_BENCH_LAUNCH_ONLY="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 2 and b"refusing to self-launch" in out.stderr and not out.stdout.strip()


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode != 0 and b"WORLD_SIZE=2" in out.stderr


def test_fine_histogram_window_arithmetic():
    """numpy twin of pf_code / pf_edge (csrc/topk_dev.h): the window code is monotone in the key, every key lies in [edge(code),
    edge(code + 1)), the feasibility window holds -lambda_min in (1e-15, 8], the obj_improve window both signs.  (The constants are
    read from the header so that the twin cannot drift.)"""
    import re
    import struct
    hdr = open(os.path.join(ROOT, "sdpcutsel_via_nn_amd", "csrc", "topk_dev.h")).read()
    PF = int(re.search(r"#define PF_BINS (\d+)", hdr).group(1))
    assert PF == 2048 and "PF_FEAS_BASE (0x10000 + ((1023 + 8) << 5) - PF_BINS)" in hdr
    assert "PF_POS_BASE (0x10000 + ((1023 - 16) << 5))" in hdr and "PF_NEG_BASE (0xFFFF - ((1023 + 16) << 5) + 1)" in hdr
    FE, PO, NE = 0x10000 + ((1023 + 8) << 5) - PF, 0x10000 + ((1023 - 16) << 5), 0xFFFF - ((1023 + 16) << 5) + 1

    def key_of(x):      # csrc/keys.h
        u = struct.unpack("<Q", struct.pack("<d", x + 0.0))[0]
        return (~u) & 0xFFFFFFFFFFFFFFFF if u >> 63 else u | (1 << 63)

    def clamp(c, n):
        return 0 if c < 0 else (n - 1 if c > n - 1 else c)

    def code(x, feas):
        c = key_of(x) >> 47
        if feas:
            return clamp(c - FE, PF)
        return PF // 2 + clamp(c - PO, PF // 2) if c >= 0x10000 else clamp(c - NE, PF // 2)

    def edge(f, feas):
        if f <= 0:
            return 0
        if feas:
            return (FE + f) << 47
        if f < PF // 2:
            return (NE + f) << 47
        return 1 << 63 if f == PF // 2 else (PO + f - PF // 2) << 47
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.normal(size=20000) * 10.0 ** rng.uniform(-8, 6, 20000),
                         [0.0, -0.0, 1e-300, -1e-300, 2.0 ** 16, 2.0 ** -16, -2.0 ** 16, -2.0 ** -16, 65535.9, -65535.9, 1e-15, 8.0]])
    for feas in (False, True):
        ks = [key_of(float(x)) for x in xs]
        cs = [code(float(x), feas) for x in xs]
        order = np.argsort(np.array(ks, dtype=np.uint64))
        assert all(cs[order[i]] <= cs[order[i + 1]] for i in range(len(order) - 1))
        for k, c in zip(ks, cs):
            assert k >= edge(c, feas) and (c == PF - 1 or k < edge(c + 1, feas))
    assert 0 < code(1e-15, True) < code(8.0, True) < PF - 1                  # the whole range of a feasibility score inside the window
    assert code(-1.0, False) < PF // 2 <= code(0.0, False) < code(1.0, False)


def test_fixed_step_lower_bound_of_the_merge_ranks():
    """numpy twin of the branch-free search of csrc/topk.hip (mergerank_body): steps 256, 128 .. 1 over the first 511 entries of a
    sorted tile of 512, one more comparison for the 512-th, give the number of entries below e -- for every position of e,
    including below the first and above the last entry -- and the ranks of all entries of several tiles, each counted in its own
    tile by position and in the others by this search, are a permutation."""
    rng = np.random.default_rng(3)
    tile = np.sort(rng.choice(10 ** 6, size=512, replace=False))

    def lower_bound(t, e):
        pos = 0
        step = 256
        while step >= 1:
            pos += step if t[pos + step - 1] < e else 0
            step >>= 1
        assert pos <= 511
        return pos + (1 if pos == 511 and t[511] < e else 0)
    for e in list(tile[:5]) + list(tile[-5:]) + [-1, 10 ** 7] + list(rng.integers(0, 10 ** 6, size=300)):
        assert lower_bound(tile, e) == int(np.searchsorted(tile, e, side="left"))
    vals = rng.permutation(5 * 512 + 100)[:5 * 512]
    tiles = [np.sort(vals[i * 512:(i + 1) * 512]) for i in range(5)]
    ranks = []
    for ti, t in enumerate(tiles):
        for p, e in enumerate(t):
            ranks.append(p + sum(lower_bound(o, e) for oi, o in enumerate(tiles) if oi != ti))
    assert sorted(ranks) == list(range(5 * 512))
    order = np.argsort(np.concatenate(tiles), kind="stable")
    assert np.array_equal(np.argsort(np.asarray(ranks), kind="stable"), order)

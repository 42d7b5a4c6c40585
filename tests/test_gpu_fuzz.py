"""Randomised differential test through the C-ABI: small random instances with mixed candidate
sizes, odd list lengths (1, 63..65, 255..257, ...), random strategies, head lengths and kernel
options, against the oracle.  Catches the corner cases the structured tests do not enumerate."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EIG_ATOL = 2e-13


@pytest.fixture(scope="module")
def lib():
    import sdpcutsel_via_nn_amd as pkg
    return pkg


def _instance(rng, n, N, kinds):
    from sdpcutsel_via_nn_amd import synthetic
    L = n * (n + 1) // 2
    Q = np.round(rng.normal(size=L) * 20) * (rng.uniform(size=L) < rng.choice([0.3, 0.7, 1.0]))
    ks = rng.choice(kinds, size=N).astype(np.int32)
    sets = np.full((N, 5), -1, dtype=np.int32)
    for k in np.unique(ks):
        m = ks == k
        sets[m, :k] = synthetic.random_index_sets(n, int(k), int(m.sum()), rng)
    return Q, sets, ks


def _point(rng, n, kind):
    iu = np.triu_indices(n)
    x = rng.uniform(0, 1, n)
    if kind == "psd":                       # X = min(x_i, x_j): lifted matrix PSD, nothing violated
        X = np.minimum(x[iu[0]], x[iu[1]])
    elif kind == "mid":                     # structured vertex: many exactly equal scores
        x = np.full(n, 0.5)
        X = rng.choice([0.0, 0.5], size=iu[0].shape[0])
    else:                                   # generic McCormick-feasible point
        lo = np.maximum(0.0, x[iu[0]] + x[iu[1]] - 1.0)
        hi = np.minimum(x[iu[0]], x[iu[1]])
        X = lo + rng.uniform(size=lo.shape[0]) * (hi - lo)
    return np.concatenate([X, x])


@pytest.mark.parametrize("seed", range(80))
def test_random_instances_against_the_oracle(lib, oracle, seed):
    from sdpcutsel_via_nn_amd import _capi, networks
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(6, 41))
    N = int(rng.choice([1, 2, 5, 63, 64, 65, 255, 256, 257, 1000, 4097, 9001]))
    kinds = [k for k in (2, 3, 4, 5) if k <= n and rng.uniform() < 0.6] or [3]
    Q, sets, ks = _instance(rng, n, N, kinds)
    vv = _point(rng, n, rng.choice(["gen", "gen", "mid", "psd"]))
    L = n * (n + 1) // 2
    sc = lib.Scorer(0)
    try:
        for k in range(2, 6):
            sc.set_network(k, *networks.load_network(k))
        sc.set_instance(n, Q)
        base = int(rng.choice([0, 0, 7, 10 ** 9]))
        sc.set_candidates(sets, ks, global_base=base)
        sc.set_option(_capi.OPT_KERNEL, int(rng.choice([_capi.KERNEL_MFMA, _capi.KERNEL_MFMA, _capi.KERNEL_VALU])))
        sc.set_option(_capi.OPT_FUSE_KEYS, int(rng.integers(0, 2)))
        sc.set_point(vv)
        sc.score(_capi.EIG | _capi.NN)
        eig, obj = sc.get_scores()
        # scores against the oracle, per candidate size
        for k in np.unique(ks):
            m = np.flatnonzero(ks == k)
            si = sets[m, :k]
            ref_obj = oracle.opt_score_batch(int(k), si, n, vv, Q)
            ref_eig = oracle.eigmin_batch(int(k), vv[L:][si], vv[:L][oracle.triu_positions(si, n)])
            assert np.abs(eig[m] - ref_eig).max() <= EIG_ATOL, (seed, k)
            tol = 1e-9 * np.maximum(np.abs(ref_obj), 1e-3 * np.abs(ref_obj).max() + 1e-12) + 1e-9
            assert np.all(np.abs(obj[m] - ref_obj) <= tol), (seed, k)
        # rankings: exact given the device's scores
        for _ in range(6):
            strat = int(rng.choice([1, 2, 4]))
            sel = int(rng.choice([0, 1, 2, max(N // 10, 1), N, N + 3, 5000]))
            max_out = int(rng.choice([sel, N, 1, 8192, 10 ** 6]))
            if strat == 4 and sel == 0:
                continue
            ids, score, total, new_strat, cnt = sc.rank(strat, sel, max_out=max_out)
            order, ref_score, ref_strat, ref_cnt = oracle.rank_arrays(strat, obj, eig, sel)
            w = min(max_out, order.shape[0])
            assert total == order.shape[0] and new_strat == ref_strat, (seed, strat, sel, max_out)
            assert np.array_equal(ids, order[:w] + base), (seed, strat, sel, max_out)
            assert np.array_equal(score, ref_score[:w] + 0.0), (seed, strat, sel, max_out)
            # the fused round returns the same head and the rows sdpcut_cut_rows gives
            sc.set_point(vv)                               # clears the scored flags (fusion option)
            r = sc.select_round(strat, sel)
            w = min(sel, order.shape[0], N)
            assert np.array_equal(r["idx"], order[:w] + base), (seed, strat, sel)
            assert np.array_equal(r["score"], ref_score[:w] + 0.0)
            assert r["n_total"] == order.shape[0] and r["new_strat"] == ref_strat
            if w:
                lam, coef, rhs, cols, kk = sc.cut_rows(order[:w])
                ld = r["coef"].shape[1]
                assert np.array_equal(r["lam"], lam) and np.array_equal(r["rhs"], rhs) and np.array_equal(r["ks"], kk)
                assert np.array_equal(r["coef"], coef[:, :ld]) and not coef[:, ld:].any()
                assert np.abs(lam - eig[order[:w]]).max() <= 1e-14
                # the same round with the cuts assembled on the device (sdpcut_round_csr, r3): the CSR block is the host
                # assembly of the padded rows above, bit for bit
                from sdpcutsel_via_nn_amd.cut_solver import rows_to_csr
                c = sc.round_csr(strat, sel, point=vv if rng.uniform() < 0.5 else None, copy=True)
                assert np.array_equal(c["idx"], r["idx"]) and np.array_equal(c["score"], r["score"]) and c["new_strat"] == r["new_strat"]
                keep = np.flatnonzero(lam < -1e-15)
                indptr, ind, val = rows_to_csr(coef[keep], cols[keep], kk[keep])
                assert np.array_equal(c["row_entry"], keep) and np.array_equal(c["indptr"], indptr), (seed, strat, sel)
                assert np.array_equal(c["indices"], ind) and np.array_equal(c["values"], val) and np.array_equal(c["rhs"], rhs[keep])
                assert np.array_equal(c["set_inds"], sets[order[:w]]) and np.array_equal(c["ks"], kk)
                # the row is v^T [[1, x^T], [x, X]] v written out for the unit eigenvector v of lam:
                # -rhs + coef . (x_rho, X_rho) = lam, and the columns are those of the index set
                for j in rng.choice(w, size=min(w, 8), replace=False):
                    c = int(order[j])
                    k = int(ks[c])
                    si = sets[c:c + 1, :k]
                    pos = oracle.triu_positions(si, n)[0]
                    assert cols[j, :k].tolist() == (L + si[0]).tolist() and cols[j, k:k + len(pos)].tolist() == pos.tolist()
                    val = -rhs[j] + coef[j, :k] @ vv[L:][si[0]] + coef[j, k:k + len(pos)] @ vv[:L][pos]
                    assert abs(val - lam[j]) <= 1e-12, (seed, c, val, lam[j])
            sc.score(_capi.EIG | _capi.NN)
    finally:
        sc.close()


@pytest.mark.parametrize("seed", range(16))
def test_rounds_that_score_for_themselves_against_the_oracle(lib, oracle, seed):
    """sdpcut_select_round on a fresh point (nothing scored): the score kernels count the leading radix digit
    of the selection keys and the selection builds its keys from the scores (no key pass).  Lists longer than
    the sort buffers, every strategy, both regimes of the combined one, generic and tie-heavy points, mixed
    candidate sizes; against the oracle's ranking of the device's own scores and against the unfused round."""
    from sdpcutsel_via_nn_amd import _capi, networks
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(8, 41))
    N = int(rng.choice([8193, 9001, 20000, 65536, 150001, 300000]))
    kinds = [k for k in (2, 3, 4, 5) if rng.uniform() < 0.5] or [3]
    Q, sets, ks = _instance(rng, n, N, kinds)
    vv = _point(rng, n, rng.choice(["gen", "gen", "mid", "psd"]))
    sc = lib.Scorer(0)
    try:
        for k in range(2, 6):
            sc.set_network(k, *networks.load_network(k))
        sc.set_instance(n, Q)
        base = int(rng.choice([0, 11, 10 ** 10]))
        sc.set_candidates(sets, ks, global_base=base)
        sc.set_point(vv)
        sc.score(_capi.EIG | _capi.NN)
        eig, obj = sc.get_scores()
        n_strong = int(((obj > 0) & (eig < -1e-15)).sum())
        for strat in (1, 2, 4):
            sels = {1, 37, 5000, 8192, int(rng.integers(1, 8193))}
            if strat == 4 and n_strong > 2:
                sels |= {max(n_strong - 1, 1), min(n_strong + 1, 8192)}      # either side of the regime switch
            for sel in sorted(sels):
                order, ref_score, ref_strat, ref_cnt = oracle.rank_arrays(strat, obj, eig, sel)
                w = min(sel, order.shape[0])
                voids = []
                for fuse in (1, 0):
                    sc.set_option(_capi.OPT_FUSE_KEYS, fuse)
                    sc.set_point(vv)                       # nothing scored: the round scores for itself
                    before = sc.get_stat(_capi.STAT_SELECT_FALLBACKS)
                    r = sc.select_round(strat, sel)
                    voids.append(sc.get_stat(_capi.STAT_SELECT_FALLBACKS) - before)
                    assert np.array_equal(r["idx"], order[:w] + base), (seed, strat, sel, fuse)
                    assert np.array_equal(r["score"], ref_score[:w] + 0.0), (seed, strat, sel, fuse)
                    assert r["n_total"] == order.shape[0] and r["new_strat"] == ref_strat, (seed, strat, sel, fuse)
                    if strat == 4:
                        assert r["counters"]["strong"] == ref_cnt["strong"] and r["counters"]["violated"] == ref_cnt["violated"]
                    if w:
                        assert np.abs(r["lam"] - eig[order[:w]]).max() <= 1e-14
                # a selection may declare itself void (a tie group of the every-entry-visited regime that the sort
                # buffers cannot hold: the full-sort path answers) -- but the same ones in both variants
                assert voids[0] == voids[1], (seed, strat, sel, voids)
    finally:
        sc.close()


@pytest.mark.parametrize("seed", range(4))
def test_scores_over_hundreds_of_binades(lib, oracle, seed):
    """Objective coefficients from 1e-120 to 1e+120 (both signs): obj_improve spreads over hundreds of binades, so
    the leading radix digit the score kernel counts takes dozens of values inside one wave (its aggregation loop
    runs out of rounds and falls back to per-lane adds) and the selection's threshold lands in arbitrary bins,
    negative keys included.  Rounds that score for themselves against the oracle's ranking of the device's scores."""
    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    rng = np.random.default_rng(7000 + seed)
    n, N = 30, int(rng.choice([20000, 70001]))
    L = n * (n + 1) // 2
    Q = np.where(rng.uniform(size=L) < 0.5, -1.0, 1.0) * 10.0 ** rng.uniform(-120, 120, size=L)
    ks = rng.choice([2, 3], size=N).astype(np.int32)
    sets = np.full((N, 5), -1, dtype=np.int32)
    for k in (2, 3):
        m = ks == k
        sets[m, :k] = synthetic.random_index_sets(n, k, int(m.sum()), rng)
    vv = _point(rng, n, "gen")
    sc = lib.Scorer(0)
    try:
        for k in (2, 3):
            sc.set_network(k, *networks.load_network(k))
        sc.set_instance(n, Q)
        sc.set_candidates(sets, ks)
        sc.set_point(vv)
        sc.score(_capi.EIG | _capi.NN)
        eig, obj = sc.get_scores()
        assert np.isfinite(obj).all()
        expo = np.frexp(np.abs(obj[obj != 0]))[1]
        assert expo.max() - expo.min() > 300           # the spread the test is about
        for strat in (2, 4, 1):
            for sel in (1, 500, 5000, 8192):
                order, ref_score, ref_strat, ref_cnt = oracle.rank_arrays(strat, obj, eig, sel)
                w = min(sel, order.shape[0])
                for fuse in (1, 0):
                    sc.set_option(_capi.OPT_FUSE_KEYS, fuse)
                    sc.set_point(vv)
                    r = sc.select_round(strat, sel)
                    assert np.array_equal(r["idx"], order[:w]), (seed, strat, sel, fuse)
                    assert np.array_equal(r["score"], ref_score[:w] + 0.0), (seed, strat, sel, fuse)
                    assert r["new_strat"] == ref_strat and r["n_total"] == order.shape[0]
    finally:
        sc.close()

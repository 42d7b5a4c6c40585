"""Synthetic workloads of BASELINE.json (SURVEY.md section 8 d).

C2: seed 7, n = 100, dense integer objective in [-50, 50] packed with the BoxQP convention of
cut_select_qp.py:318-321 (off-diagonal -q_ij, diagonal -q_ii / 2), a McCormick-feasible LP
point, and N index sets i < j < ... drawn uniformly WITH replacement (N exceeds C(100, 3)).
"""
import numpy as np

from .harness import random_mccormick_point


def make_instance(nb_vars, seed=7):
    rng = np.random.default_rng(seed)
    q = rng.integers(-50, 51, size=(nb_vars, nb_vars)).astype(np.float64)
    q = np.triu(q) + np.triu(q, 1).T                 # symmetric integer matrix
    Q = -q
    Q[np.diag_indices(nb_vars)] /= 2.0
    Q_arr = Q[np.triu_indices(nb_vars)]
    vars_values = random_mccormick_point(nb_vars, rng)
    return Q_arr, vars_values, rng


def random_index_sets(nb_vars, k, count, rng):
    """count sorted k-subsets of range(nb_vars), uniform with replacement -> int32 [count, k]."""
    out = np.empty((count, k), dtype=np.int32)
    done = 0
    while done < count:
        m = int((count - done) * 1.3) + 16
        cand = np.sort(rng.integers(0, nb_vars, size=(m, k)), axis=1)
        ok = np.all(cand[:, 1:] != cand[:, :-1], axis=1)
        cand = cand[ok][:count - done]
        out[done:done + cand.shape[0]] = cand
        done += cand.shape[0]
    return out


def make_workload(nb_vars=100, k=3, count=10 ** 6, seed=7):
    """-> dict(nb_vars, Q_arr, vars_values, set_inds [count, 5] padded with -1, ks)."""
    Q_arr, vars_values, rng = make_instance(nb_vars, seed)
    s = random_index_sets(nb_vars, k, count, rng)
    pad = np.full((count, 5), -1, dtype=np.int32)
    pad[:, :k] = s
    return dict(nb_vars=nb_vars, Q_arr=Q_arr, vars_values=vars_values, set_inds=pad,
                ks=np.full(count, k, dtype=np.int32))


# ---------------------------------------------------------------------------------------------
# C4 workload (SURVEY.md section 8 d): candidate id -> index set through Philox4x32-10.  The device
# generates the list itself (sdpcut_set_candidates_philox, csrc/philox.h); this is the numpy twin
# of the same arithmetic, used to verify sub-samples and to hand index sets to host-side checks.
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)
PHILOX_MAX_ATTEMPTS = 64


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10: counters c0..c3 (uint32 arrays), key (k0, k1) scalars -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK32 for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _PHILOX_M0 * c0, _PHILOX_M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, p1 & _MASK32, n2, p0 & _MASK32
        k0, k1 = (k0 + _PHILOX_W0) & 0xFFFFFFFF, (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def philox_index_sets(nb_vars, k, ids, seed=7):
    """Index sets of the candidates ``ids`` (any int64 array) -> int32 [len(ids), 5] padded with -1:
    k draws floor(u32 * nb_vars / 2^32), sorted, redrawn (attempt counter) until distinct."""
    ids = np.asarray(ids, dtype=np.uint64).ravel()
    out = np.full((ids.shape[0], 5), -1, dtype=np.int32)
    todo = np.arange(ids.shape[0])
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for attempt in range(PHILOX_MAX_ATTEMPTS):
        if not todo.size:
            break
        lo, hi = ids[todo] & _MASK32, ids[todo] >> np.uint64(32)
        a = np.full(todo.shape[0], attempt, dtype=np.uint64)
        r = list(philox4x32_10(lo, hi, a, np.zeros_like(a), k0, k1))
        if k > 4:
            r += list(philox4x32_10(lo, hi, a, np.ones_like(a), k0, k1))
        v = np.stack([(x.astype(np.uint64) * np.uint64(nb_vars)) >> np.uint64(32) for x in r[:k]], axis=1).astype(np.int32)
        v.sort(axis=1)
        ok = np.all(v[:, 1:] != v[:, :-1], axis=1)
        out[todo[ok], :k] = v[ok]
        todo = todo[~ok]
    out[todo, :k] = np.arange(k, dtype=np.int32)
    return out


def make_philox_workload(nb_vars=1000, k=3, seed=7):
    """Instance part of the C4 workload (objective table and LP point as in :func:`make_instance`);
    the index sets live on the device -> dict(nb_vars, Q_arr, vars_values, k, seed)."""
    Q_arr, vars_values, _ = make_instance(nb_vars, seed)
    return dict(nb_vars=nb_vars, Q_arr=Q_arr, vars_values=vars_values, k=k, seed=seed)

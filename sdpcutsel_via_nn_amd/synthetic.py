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

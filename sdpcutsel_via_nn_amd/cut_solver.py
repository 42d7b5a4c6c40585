"""Host-side mirror of the reference's cut-selection surface, backed by the GPU library.

The cutting-plane loop of the reference (cut_select_qp.py:149-200, cut_select_qcqp.py:63-112)
talks to the hot path through four methods of ``CutSolver``:

    _load_neural_nets()                                               cut_select_qp.py:284-303
    _sel_eigcut_by_ordering_on_measure(strat, vars_values, cut_round, sel_size=0)   :543-703
    _gen_eigcuts_selected(strat, sel_size, rank_list, strong_only=False, vars_values=None)  :705-755
    _get_eigendecomp(dim_subpr, curr_pt, X_slice, ev_yes)             :788-797

:class:`GpuCutSelectionMixin` provides exactly these, with the same arguments, return layouts
and ordering, reading the same instance state (``self._agg_list``, ``self._nb_lifted``,
``self._Q_arr``, ``self._nb_vars``, ``self._dim``, ``self._my_prob``).  Put it in front of the
reference class to drop it into the unmodified loop:

    class GpuCutSolver(GpuCutSelectionMixin, cut_select_qp.CutSolver): pass

All arithmetic (gather, eigen-decomposition, MLP, ranking, cut coefficients) runs in
libsdpcut_hip.so; this file only marshals arrays and builds the Python objects the loop
expects.  No CPU fallback exists: without the library / a gfx950 GPU the methods raise.
"""
from collections.abc import Sequence

import numpy as np

from . import _capi, networks

_THRES_NEG_EIGVAL = -10 ** (-15)      # cut_select_qp.py:24
_BIG_M = 1000                         # cut_select_qp.py:26
_HEAD = 5000                          # _SDP_CUTS_PER_ROUND_MAX (:37): rank-list head fetched eagerly


def _default_sparse_pair():
    try:
        import cplex                      # the reference's LP object, if installed
        return cplex.SparsePair
    except ImportError:
        from .harness import SparsePair
        return SparsePair


def rows_to_csr(coef, cols, ks):
    """Padded cut rows (stride SDPCUT_ROW_LD, row c has k_c(k_c+3)/2 live entries) -> CSR
    (indptr int64 [R+1], indices int64, values float64)."""
    ks = np.asarray(ks, dtype=np.int64)
    lens = ks * (ks + 3) // 2
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    live = np.arange(coef.shape[1])[None, :] < lens[:, None]
    return indptr, np.ascontiguousarray(cols[live], dtype=np.int64), np.ascontiguousarray(coef[live], dtype=np.float64)


class FeasEntry(tuple):
    """``(set_inds, -eigval, Xarr_inds, dim_act)`` of cut_select_qp.py:649, remembering which
    candidate it came from so that cut generation needs no search."""
    agg_idx = -1
    binding = None

    def __new__(cls, items, agg_idx, binding=None):
        self = super().__new__(cls, items)
        self.agg_idx = agg_idx
        self.binding = binding
        return self


class RankList(Sequence):
    """Lazy stand-in for the reference's sorted ``rank_list``.

    The reference materialises N Python tuples; downstream code only looks at the first
    ``sel_size`` of them.  This object keeps the ranking on the device, fetches windows on
    demand (``sdpcut_rank_fetch``) and builds entry tuples only for what is indexed.
    ``a + b`` (cut_select_qcqp.py:79) and slicing return plain lists of entries.
    """

    def __init__(self, owner, binding, kind, total, vars_values, head_idx, head_score, strat=None, sel_size=0):
        self._owner, self._b, self._kind, self._n = owner, binding, kind, int(total)
        self._vv = vars_values
        self._strat, self._sel_size = strat, sel_size
        self._idx, self._score = head_idx, head_score
        self._have = head_idx.shape[0]
        self._point = binding.point_token

    # -- data access -----------------------------------------------------------------
    def _need(self, upto):
        """The head came from the device's top-k selection; anything beyond it asks for the
        complete ranking (full device sort), once."""
        upto = min(upto, self._n)
        if upto <= self._have:
            return
        if self._b.point_token != self._point:
            raise RuntimeError("rank list is stale: the device now holds the scores of a newer LP point")
        idx, sc, total, _, _ = self._b.scorer.rank(self._strat, self._sel_size, max_out=self._n)
        self._idx, self._score, self._have = idx, sc, idx.shape[0]

    def ids(self, count=None):
        count = self._n if count is None else min(count, self._n)
        self._need(count)
        return self._idx[:count]

    def scores(self, count=None):
        count = self._n if count is None else min(count, self._n)
        self._need(count)
        return self._score[:count]

    def _entry(self, pos):
        idx, score = int(self._idx[pos]), float(self._score[pos])
        set_inds, Xarr_inds = self._b.agg_entry(idx)
        if self._kind == 1:
            return FeasEntry((set_inds, score, Xarr_inds, len(set_inds)), idx, self._b)
        L = self._b.nb_lifted
        curr_pt = tuple(self._vv[L + i] for i in set_inds)
        X_slice = tuple(self._vv[i] for i in Xarr_inds)
        return (idx, score, curr_pt, X_slice)

    # -- Sequence protocol -----------------------------------------------------------
    def __len__(self):
        return self._n

    def __getitem__(self, key):
        if isinstance(key, slice):
            rng = range(*key.indices(self._n))
            if len(rng):
                self._need(max(rng) + 1)
            return [self._entry(p) for p in rng]
        if key < 0:
            key += self._n
        if not 0 <= key < self._n:
            raise IndexError(key)
        self._need(key + 1)
        return self._entry(key)

    def __iter__(self):
        for lo in range(0, self._n, 65536):
            self._need(min(lo + 65536, self._n))
            for p in range(lo, min(lo + 65536, self._n)):
                yield self._entry(p)

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)


class _Binding(object):
    """Device-side twin of one ``agg_list``: a Scorer handle holding its index sets."""

    def __init__(self, scorer, agg_list, nb_vars, nb_lifted):
        self.scorer, self.agg_list = scorer, agg_list
        self.nb_vars, self.nb_lifted = nb_vars, nb_lifted
        self.rank_serial = 0
        self.point_token = None
        self.scored = 0
        self.set_arr = None      # [N, 5] when bound from arrays

    def agg_entry(self, idx):
        if self.agg_list is not None:
            e = self.agg_list[idx]
            return e[0], e[1]
        k = int(self.ks[idx])
        s = [int(v) for v in self.set_arr[idx, :k]]
        n = self.nb_vars
        pos = [n * s[a] - s[a] * (s[a] + 1) // 2 + s[b] for a in range(k) for b in range(a, k)]
        return s, pos


class AggArrays(Sequence):
    """Array-backed ``agg_list``: the candidate records of cut_select_qp.py:529-540 built on
    demand.  A Python list of 1.7e6 tuples (spar125-075-1, dim 4) takes minutes to build and
    GBs to hold; the loop only ever reads ``[0]`` / ``[1]`` of the few thousand selected entries."""

    def __init__(self, set_inds, ks, nb_vars, Q_arr=None):
        self.set_inds = np.ascontiguousarray(set_inds, dtype=np.int32)
        self.ks = np.ascontiguousarray(ks, dtype=np.int32)
        self.nb_vars, self.Q_arr = nb_vars, Q_arr

    def __len__(self):
        return self.ks.shape[0]

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return [self[i] for i in range(*idx.indices(len(self)))]
        k = int(self.ks[idx])
        s = [int(v) for v in self.set_inds[idx, :k]]
        n = self.nb_vars
        pos = [n * s[a] - s[a] * (s[a] + 1) // 2 + s[b] for a in range(k) for b in range(a, k)]
        if self.Q_arr is None:
            return (s, pos, None, None)
        q = [self.Q_arr[p] for p in pos]
        max_elem = k * abs(max(q, key=abs))
        max_elem += 1 if not max_elem else 0
        return (s, pos, tuple(np.divide(q, max_elem)), max_elem)


class GpuCutSelectionMixin(object):
    """The four hot-path methods of the reference's ``CutSolver`` on the GPU."""

    _gpu_device = 0
    _sparse_pair = None

    # ------------------------------------------------------------------ a11 loader
    def _load_neural_nets(self):
        """Create the GPU handle pool and upload the MLPs for d = 2..self._dim
        (replaces the ctypes load of NNs.so, cut_select_qp.py:284-303)."""
        self._gpu_nets = {}
        for d in range(2, self._dim + 1):
            self._gpu_nets[d] = networks.load_network(d)
        # one entry per dimension like the reference's (func, buffer) list, for code that
        # only checks its length
        self._nns = [(None, None)] * (self._dim - 1)
        self._gpu_bindings = {}

    # ------------------------------------------------------------------ device binding
    def _gpu_new_scorer(self):
        sc = _capi.Scorer(self._gpu_device)
        for d, (widths, params) in getattr(self, "_gpu_nets", {}).items():
            sc.set_network(d, widths, params)
        return sc

    def _gpu_bind(self):
        """Scorer bound to the current ``self._agg_list`` (cached per list object: the QCQP
        loop swaps two lists every round, cut_select_qcqp.py:75-78)."""
        if not hasattr(self, "_gpu_bindings") or self._gpu_bindings is None:
            self._gpu_bindings = {}
        if not hasattr(self, "_gpu_nets"):
            self._gpu_nets = {}
        agg = self._agg_list
        key = id(agg)
        b = self._gpu_bindings.get(key)
        if b is not None and b.agg_list is agg and b.n_at_bind == len(agg):
            return b
        N = len(agg)
        if isinstance(agg, AggArrays):          # enumerated on our side: the arrays already exist
            S, ks = agg.set_inds, agg.ks
        else:
            S = np.full((max(N, 1), 5), -1, dtype=np.int32)
            ks = np.zeros(max(N, 1), dtype=np.int32)
            for i, e in enumerate(agg):
                s = e[0]
                ks[i] = len(s)
                S[i, :len(s)] = s
        sc = self._gpu_new_scorer()
        sc.set_instance(self._nb_vars, np.asarray(self._Q_arr, dtype=np.float64))
        sc.set_candidates(S[:N], ks[:N])
        b = _Binding(sc, agg, self._nb_vars, self._nb_lifted)
        b.n_at_bind = N
        b.ks = ks[:N]
        self._gpu_bindings[key] = b
        return b

    def gpu_bind_arrays(self, set_inds, ks, nb_vars, Q_arr, global_base=0):
        """Array entry (no Python agg_list): used for large synthetic / enumerated covers."""
        if not hasattr(self, "_gpu_nets"):
            self._gpu_nets = {}
        sc = self._gpu_new_scorer()
        sc.set_instance(nb_vars, Q_arr)
        sc.set_candidates(set_inds, ks, global_base)
        b = _Binding(sc, None, nb_vars, nb_vars * (nb_vars + 1) // 2)
        b.set_arr, b.ks, b.n_at_bind = np.asarray(set_inds), np.asarray(ks), len(ks)
        return b

    @staticmethod
    def _gpu_point(b, vars_values, flags):
        vv = np.ascontiguousarray(vars_values, dtype=np.float64)
        token = (vv.ctypes.data, vv.shape[0], hash(vv.tobytes()))
        if b.point_token != token:
            b.scorer.set_point(vv)
            b.point_token, b.scored = token, 0
        need = flags & ~b.scored
        if need:
            b.scorer.score(need)
            b.scored |= need
        return vv

    # ------------------------------------------------------------------ a7-a9 selection
    def _sel_eigcut_by_ordering_on_measure(self, strat, vars_values, cut_round, sel_size=0):
        """Strategies 1 (feasibility), 2 (optimality via MLP), 4 (combined), 5 (random);
        same returns as cut_select_qp.py:543-703.  Strategies 3 / -1 need an exact SDP
        solver per candidate and are out of scope."""
        if strat == 5:
            np.random.shuffle(self._agg_list)      # in place, like :636
            return self._agg_list
        if strat not in (1, 2, 4):
            raise NotImplementedError("exact-SDP strategies (3, -1) are not part of the GPU path")
        b = self._gpu_bind()
        N = len(self._agg_list)
        sel_size = min(sel_size, N)
        flags = {1: _capi.EIG, 2: _capi.NN, 4: _capi.EIG | _capi.NN}[strat]
        if N == 0:
            return []       # strat 4 included: sel_size is clamped to 0 and the reference falls through
        vv = self._gpu_point(b, vars_values, flags)
        # head fetched eagerly: what the loop can consume (sel_size is only passed for strat 4;
        # for 1 / 2 the cap of :37 bounds it).  Heads <= 8192 take the device's top-k select path.
        head = min(N, sel_size if (strat == 4 and sel_size > 0) else _HEAD)
        idx, score, total, new_strat, counters = b.scorer.rank(strat, sel_size, head)
        b.rank_serial += 1
        rl = RankList(self, b, 1 if strat == 1 else 2, total, vv, idx, score, strat=strat, sel_size=sel_size)
        rl.counters = counters
        if strat == 4:
            # the reference divides by sel_size and swallows the ZeroDivisionError, falling
            # through to `return rank_list` (cut_select_qp.py:629-632, 703)
            return rl if sel_size == 0 else (new_strat, rl)
        return rl

    # ------------------------------------------------------------------ a10 generation
    def _gen_eigcuts_selected(self, strat, sel_size, rank_list, strong_only=False, vars_values=None):
        """Eigen-cuts of the first ``sel_size`` ranked candidates, appended to
        ``self._my_prob.linear_constraints`` (cut_select_qp.py:705-755)."""
        sel_size = min(sel_size, len(rank_list))
        opt_sel, feas_sel = strat in (2, 3, 4, -1), strat == 1
        pair = self._sparse_pair or _default_sparse_pair()
        rows, rhs_out = [], []
        if sel_size > 0:
            b, idx, vv = None, None, vars_values
            if isinstance(rank_list, RankList):
                b, idx, vv = rank_list._b, rank_list.ids(sel_size), rank_list._vv
                if opt_sel and strong_only:                       # :725-726
                    stop = np.nonzero(rank_list.scores(sel_size) <= 0)[0]
                    if stop.size:
                        idx = idx[:stop[0]]
            else:
                entries = list(rank_list[0:sel_size])
                if opt_sel:
                    if strong_only:
                        cut = next((p for p, e in enumerate(entries) if e[1] <= 0), len(entries))
                        entries = entries[:cut]
                    idx = np.array([e[0] for e in entries], dtype=np.int64)
                    b = self._gpu_bind()
                elif feas_sel and all(isinstance(e, FeasEntry) for e in entries):
                    idx = np.array([e.agg_idx for e in entries], dtype=np.int64)
                    b = self._find_binding_for(entries)
                if b is None or idx is None:
                    return self._gen_from_entries(entries, feas_sel, vars_values, pair)
            if idx.size:
                if not opt_sel or vv is None:
                    vv = vars_values
                self._gpu_point(b, vv, 0)
                lam, coef, rhs, cols, ks = b.scorer.cut_rows(idx - b.scorer.base)
                keep = np.nonzero(lam < _THRES_NEG_EIGVAL)[0]          # :743
                store = self._my_prob.linear_constraints
                if hasattr(store, "add_csr"):
                    # batched assembly (SURVEY 8 f row 4): the padded device rows become one CSR
                    # block with array operations, no Python object per cut
                    indptr, ind, val = rows_to_csr(coef[keep], cols[keep], ks[keep])
                    store.add_csr(indptr, ind, val, rhs[keep], "G")
                    return int(keep.size)
                for c in keep:
                    w = int(ks[c]) * (int(ks[c]) + 3) // 2
                    rows.append(pair(ind=cols[c, :w].tolist(), val=coef[c, :w].tolist()))
                    rhs_out.append(float(rhs[c]))
        self._my_prob.linear_constraints.add(lin_expr=rows, rhs=rhs_out, senses=["G"] * len(rows))
        return len(rows)

    def _find_binding_for(self, entries):
        """Binding whose candidate list ALL the FeasEntry objects index, else None (the QCQP
        feasibility-only round mixes entries of two lists, cut_select_qcqp.py:79)."""
        b = entries[0].binding
        if b is not None and all(e.binding is b for e in entries):
            return b
        return None

    def _gen_from_entries(self, entries, feas_sel, vars_values, pair):
        """Generic path for entries that do not carry a candidate index (random strategy, foreign
        lists): host gather of the tiny slices, GPU batched eigen-decomposition, row assembly."""
        L = self._nb_lifted
        vv = np.asarray(vars_values, dtype=np.float64)
        rows, rhs_out = [None] * len(entries), [None] * len(entries)
        by_k = {}
        for p, e in enumerate(entries):
            set_inds, Xarr_inds = (e[0], e[2]) if feas_sel else (e[0], e[1])
            by_k.setdefault(len(set_inds), []).append((p, set_inds, Xarr_inds))
        sc = self._gpu_any_scorer()
        for k, items in by_k.items():
            x = np.array([[vv[L + i] for i in it[1]] for it in items])
            X = np.array([[vv[i] for i in it[2]] for it in items])
            w, v = sc.eig_batch(k, x, X, want_vectors=True)
            for (p, set_inds, Xarr_inds), lam, vec in zip(items, w[:, 0], v[:, :, 0]):
                if lam < _THRES_NEG_EIGVAL:
                    ev = np.where(abs(vec) <= -_THRES_NEG_EIGVAL, 0, vec)
                    coef = [ev[a] * ev[c] * 2 if a != c else ev[a] * ev[c]
                            for a in range(k + 1) for c in range(max(a, 1), k + 1)]
                    rows[p] = pair(ind=[i + L for i in set_inds] + list(Xarr_inds), val=[float(t) for t in coef])
                    rhs_out[p] = float(-ev[0] * ev[0])
        rows_f = [r for r in rows if r is not None]
        rhs_f = [r for r in rhs_out if r is not None]
        self._my_prob.linear_constraints.add(lin_expr=rows_f, rhs=rhs_f, senses=["G"] * len(rows_f))
        return len(rows_f)

    def _gpu_any_scorer(self):
        for b in getattr(self, "_gpu_bindings", {}).values():
            return b.scorer
        if getattr(self, "_gpu_aux_scorer", None) is None:
            self._gpu_aux_scorer = _capi.Scorer(self._gpu_device)
        return self._gpu_aux_scorer

    # ------------------------------------------------------------------ triangle inequalities (8 f row 3)
    _THRES_TRI_VIOL = 10 ** (-7)
    _TRI_CUTS_PER_ROUND_MIN = 5000
    _TRI_CUTS_PER_ROUND_MAX = 10000

    def _gpu_dense_adj(self):
        """``self._Q_adj`` as a dense boolean array (the reference keeps a cvxopt spmatrix)."""
        adj = self._Q_adj
        if hasattr(adj, "I") and hasattr(adj, "J"):            # cvxopt.spmatrix
            out = np.zeros((self._nb_vars, self._nb_vars), dtype=bool)
            out[np.array(list(adj.I), dtype=int), np.array(list(adj.J), dtype=int)] = True
            return out
        if hasattr(adj, "a"):                                   # dense stand-in used by make_golden.py
            return np.asarray(adj.a) != 0
        return np.asarray(adj) != 0

    def _preprocess_triangle_ineq(self):
        """Triples to consider as triangle inequalities (cut_select_qp.py:799-822); the list
        lives on the device, the host keeps a copy for building the rows."""
        sc = _capi.Scorer(self._gpu_device)
        sc.set_instance(self._nb_vars, np.asarray(self._Q_arr, dtype=np.float64))
        self._gpu_tri = sc
        self._gpu_tri_triples, self._gpu_tri_density = sc.tri_preprocess(self._gpu_dense_adj())
        # the reference's bookkeeping lists, for code that only looks at their length
        self._idx_list_tri = self._gpu_tri_triples
        self._rank_list_tri = None

    def _separate_and_add_triangle(self, sel_size, vars_values):
        """Separate the violated triangle inequalities at the current point, rank them by
        (density, violation) and append the selected ones (cut_select_qp.py:824-863)."""
        sc, tri, L, n = self._gpu_tri, self._gpu_tri_triples, self._nb_lifted, self._nb_vars
        sc.set_point(np.ascontiguousarray(vars_values, dtype=np.float64))
        ent, vio, nb_viol = sc.tri_separate(self._TRI_CUTS_PER_ROUND_MAX)
        nb_tri_cuts = max(min(self._TRI_CUTS_PER_ROUND_MIN, int(np.floor(sel_size * nb_viol))),
                          min(self._TRI_CUTS_PER_ROUND_MAX, nb_viol))                       # :844-845
        pair = self._sparse_pair or _default_sparse_pair()
        coeffs = {0: [-1, -1, 1, 1], 1: [-1, 1, -1, 1], 2: [1, -1, -1, 1], 3: [1, 1, 1, -1, -1, -1]}
        rows, rhs = [], []
        for e in ent[:nb_tri_cuts]:
            t, c = divmod(int(e), 4)
            a, b, d = (int(v) for v in tri[t])
            ra, rb = n * a - a * (a + 1) // 2, n * b - b * (b + 1) // 2
            X = [ra + b, ra + d, rb + d]                        # Xarr_inds[1], [2], [4]
            if c == 3:
                rows.append(pair(ind=X + [a + L, b + L, d + L], val=coeffs[3]))
                rhs.append(-1)
            else:
                rows.append(pair(ind=X + [(a, b, d)[c] + L], val=coeffs[c]))
                rhs.append(0)
        self._my_prob.linear_constraints.add(lin_expr=rows, rhs=rhs, senses=["G"] * len(rows))
        return len(rows)

    # the reference reaches these two through name-mangled private names inside class CutSolver
    _CutSolver__preprocess_triangle_ineq = _preprocess_triangle_ineq
    _CutSolver__separate_and_add_triangle = _separate_and_add_triangle

    # ------------------------------------------------------------------ a6 eigen helper
    def _get_eigendecomp(self, dim_subpr, curr_pt, X_slice, ev_yes):
        """Eigen-decomposition of [[1, x^T],[x, X]] (cut_select_qp.py:788-797): ascending
        eigenvalues, and eigenvectors as columns when ``ev_yes`` (numpy's eigh layout)."""
        sc = self._gpu_any_scorer()
        x = np.asarray(curr_pt, dtype=np.float64)[None, :]
        X = np.asarray(X_slice, dtype=np.float64)[None, :]
        if ev_yes:
            w, v = sc.eig_batch(dim_subpr, x, X, want_vectors=True)
            return w[0], v[0]
        return sc.eig_batch(dim_subpr, x, X)[0]


class CutSolver(GpuCutSelectionMixin):
    """Stand-alone twin of the reference's ``CutSolver`` state (cut_select_qp.py:43-71) for use
    without the reference installed: holds the instance tables and the candidate list, and
    exposes the four GPU-backed hot-path methods.  The constants keep the reference's names."""
    _THRES_MIN_OPT = 0
    _THRES_NEG_EIGVAL = _THRES_NEG_EIGVAL
    _BIG_M = _BIG_M
    _SDP_CUTS_PER_ROUND_MAX = 5000
    _THRES_MAX_SUBS = 4 * (10 ** 6)

    def __init__(self, device=0):
        self._gpu_device = device
        self._dim = 0
        self._nb_vars = 0
        self._nb_lifted = 0
        self._Q_arr = []
        self._my_prob = None
        self._agg_list = []
        self._nns = None

    def set_instance(self, nb_vars, Q_arr, agg_list, dim, my_prob=None):
        """Bind an instance: packed objective, candidate records (reference layout, only
        ``[0]`` = set_inds and ``[1]`` = Xarr_inds of each record are read) and the LP object."""
        assert dim <= 5, "Keep SDP vertex cover low-dimensional (<=5)!"      # cut_select_qp.py:93
        self._nb_vars, self._nb_lifted = nb_vars, nb_vars * (nb_vars + 1) // 2
        self._Q_arr, self._agg_list, self._dim = Q_arr, agg_list, dim
        self._my_prob = my_prob
        self._load_neural_nets()

    @staticmethod
    def selection_size(sel_size, nb_subprobs, minimum=0):
        """sel_size rule of cut_select_qp.py:123-125 (QCQP adds `minimum=1`, cut_select_qcqp.py:57-58)."""
        assert 0 < sel_size, "The selection size must be a % or number (of cuts) >0!"
        s = min(int(np.floor(sel_size * nb_subprobs)) if sel_size < 1 else min(sel_size, nb_subprobs),
                CutSolver._SDP_CUTS_PER_ROUND_MAX)
        return max(s, minimum)

    # ------------------------------------------------------------------ round harness (SURVEY 8 f row 2)
    _CONVERGENCE_TOL = 10 ** (-3)         # cut_select_qp.py:29

    def cut_select_algo(self, filename, dim, sel_size, strat=2, nb_rounds_cuts=20, term_on=False,
                        triangle_on=False, strong_only=False):
        """Algorithm 1 on a BoxQP ``.in`` file without CPLEX: the call sequence of
        cut_select_qp.py:73-221 (parse -> vertex cover -> McCormick relaxation -> rounds of
        [select, generate, re-solve]) with scipy's HiGHS as LP solver, our C++ cover
        enumeration and the GPU-backed selection / triangle separation.  Dense cuts (strat 0),
        exact-SDP strategies and chordal extensions are out of scope.
        Returns the reference's default tuple
        (bounds per round, total time, round times, separation times, nb cuts per round, [], nb candidates)."""
        from timeit import default_timer as timer
        from . import harness
        assert strat in (1, 2, 4, 5), "strategies on the GPU path: 1 feasibility, 2 optimality, 4 combined, 5 random"
        assert 0 < sel_size, "The selection size must be a % or number (of cuts) >0!"
        assert dim <= 5, "Keep SDP vertex cover low-dimensional (<=5)!"
        time_begin = timer()
        nbs_sdp_cuts, nbs_tri_cuts, curr_obj_vals, round_times, sep_times = [0], [], [], [], []
        self._dim = dim
        if strat in (2, 4):
            self._load_neural_nets()
        inst = harness.parse_boxqp(filename)
        self._nb_vars, self._nb_lifted, self._Q_arr = inst["nb_vars"], inst["nb_lifted"], inst["Q_arr"]
        self._Q_adj = inst["adj"]
        self._my_prob = my_prob = harness.LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
        set_inds, ks, nb_subprobs = _capi.enumerate_cover(inst["adj"], dim, max_subs=self._THRES_MAX_SUBS)
        if nb_subprobs >= self._THRES_MAX_SUBS or nb_rounds_cuts == 0:          # :117-120
            return [0, 0], timer() - time_begin, 0, 0, [0], 0, nb_subprobs
        self._agg_list = AggArrays(set_inds, ks, self._nb_vars, self._Q_arr)
        sel_size_frac = sel_size
        sel_size = self.selection_size(sel_size, nb_subprobs)
        t0 = timer()
        my_prob.linear_constraints.add(*harness.mccormick_rows(self._nb_vars, inst["adj"]))
        sep_times.append(timer() - t0)
        t0 = timer()
        my_prob.solve()
        round_times.append(timer() - t0 + sep_times[0])
        curr_obj_vals.append(my_prob.get_objective_value())
        vars_values = np.array(my_prob.get_values())
        if triangle_on:
            self._preprocess_triangle_ineq()                                       # :140-141
        strat_change = strat
        for cut_round in range(1, nb_rounds_cuts + 1):
            if (term_on and len(curr_obj_vals) >= 3 and curr_obj_vals[-1] != curr_obj_vals[0] and
                    (curr_obj_vals[-1] - curr_obj_vals[-2]) / (curr_obj_vals[-1] - curr_obj_vals[0])
                    < self._CONVERGENCE_TOL):
                break                                                              # :153-156
            t_sep = timer()
            if strat == 4:
                res = self._sel_eigcut_by_ordering_on_measure(strat, vars_values, cut_round, sel_size=sel_size)
                strat_change, rank_list = res if isinstance(res, tuple) else (strat, res)
            else:
                rank_list = self._sel_eigcut_by_ordering_on_measure(strat, vars_values, cut_round)
            nbs_sdp_cuts.append(self._gen_eigcuts_selected(strat, sel_size, rank_list, strong_only=strong_only,
                                                           vars_values=vars_values))
            nbs_tri_cuts.append(self._separate_and_add_triangle(sel_size_frac, vars_values) if triangle_on else 0)
            sep_times.append(timer() - t_sep)
            strat = strat_change                                                   # :188
            t0 = timer()
            my_prob.solve()
            round_times.append(timer() - t0 + sep_times[-1])
            curr_obj_vals.append(my_prob.get_objective_value())
            vars_values = np.array(my_prob.get_values()).astype(float)
        return ([-obj for obj in curr_obj_vals], timer() - time_begin, round_times, sep_times, nbs_sdp_cuts,
                nbs_tri_cuts, nb_subprobs)


class CutSolverQCQP(CutSolver):
    """QCQP composition of the path (cut_select_qcqp.py:63-103): the objective cover is ranked
    with ``strat``, the constraint-only cover with feasibility, the lists are concatenated."""

    def select_and_generate_round(self, strat, vars_values, cut_round, sel_size, agg_list, agg_list_cons):
        """One round of cut_select_qcqp.py:64-98.  Returns
        (new_strat, rank_list, nb_sdp_cuts, nb_opt_cuts)."""
        strat_old = strat
        self._agg_list = agg_list
        if strat == 5:
            rank_list = self._sel_eigcut_by_ordering_on_measure(strat, vars_values, cut_round)
            return strat, rank_list, self._gen_eigcuts_selected(strat, sel_size, rank_list,
                                                                vars_values=vars_values), 0
        if strat == 4:
            strat, comb_obj = self._sel_eigcut_by_ordering_on_measure(strat, vars_values, cut_round,
                                                                      sel_size=sel_size)
        else:
            comb_obj = self._sel_eigcut_by_ordering_on_measure(strat, vars_values, cut_round)
        self._agg_list = agg_list_cons                       # :75
        feas_cons = self._sel_eigcut_by_ordering_on_measure(1, vars_values, cut_round)
        self._agg_list = agg_list                            # :78
        n_obj = min(len(comb_obj), sel_size)
        rank_list = comb_obj[0:n_obj] + feas_cons[0:sel_size - n_obj]      # == (A + B)[0:sel_size], :79
        if strat_old == 1:
            nb = self._gen_eigcuts_selected(strat_old, sel_size, rank_list, vars_values=vars_values)
            return strat, rank_list, nb, 0
        # :85-92 counters, from the device arrays instead of a Python loop over N tuples
        nb_opt_cuts = int(np.count_nonzero(comb_obj.scores() > _BIG_M)) if len(comb_obj) else 0
        nb_cuts_combined = sum(1 for e in rank_list if isinstance(e[0], int))
        rest = sel_size - nb_cuts_combined
        nb_a = self._gen_eigcuts_selected(1, rest, feas_cons[0:rest], vars_values=vars_values)
        nb_b = self._gen_eigcuts_selected(strat_old, nb_cuts_combined, comb_obj[0:nb_cuts_combined],
                                          vars_values=vars_values)
        return strat, rank_list, nb_a + nb_b, nb_opt_cuts

    def cut_select_algo(self, filename, dim, sel_size=0.1, strat=2, nb_rounds_cuts=20):
        """Algorithm 1 adapted to QCQP on an OSiL file without CPLEX: the call sequence of
        cut_select_qcqp.py:16-113 (parse -> McCormick on the objective's edges -> LP -> two
        covers -> rounds) with HiGHS, our cover enumeration and the GPU-backed selection.
        Returns (objective value per round, sel_size, nb cuts per round, nb optimality cuts per round)."""
        from . import harness
        assert strat in (1, 2, 4, 5), "strategies on the GPU path: 1 feasibility, 2 optimality, 4 combined, 5 random"
        assert 0 < sel_size, "The selection size must be a % or number (of cuts) >0!"
        assert dim <= 5, "Keep SDP vertex cover low-dimensional (<=5)!"
        self._dim = dim
        inst = harness.parse_osil(filename)
        self._nb_vars, self._nb_lifted, self._Q_arr = inst["nb_vars"], inst["nb_lifted"], inst["Q_arr"]
        self._Q_adj, self._Q_adj_cons = inst["adj"], inst["adj_cons"]
        self._my_prob = my_prob = harness.LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
        my_prob.linear_constraints.add(inst["rows"], inst["rhs"], inst["senses"])
        my_prob.linear_constraints.add(*harness.mccormick_rows(self._nb_vars, inst["adj"]))      # :36
        self._load_neural_nets()
        my_prob.solve()
        obj_values_rounds = [my_prob.get_objective_value()]
        vars_values = np.array(my_prob.get_values())
        (So, ko), (Sc, kc) = harness.qcqp_covers(inst, dim, _capi.enumerate_cover)               # :50, :314-334
        agg_list = AggArrays(So, ko, self._nb_vars, self._Q_arr)
        agg_list_cons = AggArrays(Sc, kc, self._nb_vars, self._Q_arr)
        self._agg_list = agg_list
        sel_size = self.selection_size(sel_size, len(agg_list), minimum=1)                       # :55-58
        nbs_opt_cuts = [0] * (nb_rounds_cuts + 1)
        nbs_sdp_cuts = [0]
        for cut_round in range(1, nb_rounds_cuts + 1):
            strat, rank_list, nb_sdp_cuts, nb_opt = self.select_and_generate_round(
                strat, vars_values, cut_round, sel_size, agg_list, agg_list_cons)
            nbs_opt_cuts[cut_round] = nb_opt
            nbs_sdp_cuts.append(nb_sdp_cuts)
            my_prob.solve()
            obj_values_rounds.append(my_prob.get_objective_value())
            vars_values = np.array(my_prob.get_values()).astype(float)
        return obj_values_rounds, sel_size, nbs_sdp_cuts, nbs_opt_cuts

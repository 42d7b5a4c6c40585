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
_FUSED_HEAD_MAX = 16384               # longest head the fused round (sdpcut_round_csr) assembles


_SPARSE_PAIR = None


def _default_sparse_pair():
    """cplex.SparsePair if the reference's LP library is installed, else the stand-alone twin; looked up once
    (a failing ``import cplex`` walks the whole module path: ~60 us per call, once per round before it was cached)"""
    global _SPARSE_PAIR
    if _SPARSE_PAIR is None:
        try:
            import cplex                      # the reference's LP object, if installed
            _SPARSE_PAIR = cplex.SparsePair
        except ImportError:
            from .harness import SparsePair
            _SPARSE_PAIR = SparsePair
    return _SPARSE_PAIR


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

    def __init__(self, owner, binding, kind, total, vars_values, head_idx, head_score, strat=None, sel_size=0, fused=None):
        self._owner, self._b, self._kind, self._n = owner, binding, kind, int(total)
        self._vv = vars_values
        self._strat, self._sel_size = strat, sel_size
        self._idx, self._score = head_idx, head_score
        self._have = head_idx.shape[0]
        self._point = binding.point_token
        # fused round (sdpcut_round_csr): the cuts of the head already sit, assembled, in the scorer's pinned block;
        # `fused` holds views of it, good until the scorer runs its next round (round_count moves on)
        self._fused, self._fused_at = fused, binding.scorer.round_count
        self._head_sets = self._head_ks = None
        if fused is not None:
            self._head_sets, self._head_ks = fused["set_inds"].copy(), fused["ks"].copy()

    def fused_rows(self, count, strong_only=False, vars_values=None):
        """The assembled cuts of the first ``count`` entries if the fused round's block still holds them:
        (indptr, indices, values, rhs) views, or None (stale block, longer head, another LP point)."""
        f = self._fused
        if f is None or self._b.scorer.round_count != self._fused_at or count > self._have:
            return None
        if vars_values is not None and vars_values is not self._vv and not np.array_equal(vars_values, self._vv):
            return None
        if strong_only:                                   # cut_select_qp.py:725-726
            stop = np.flatnonzero(self._score[:count] <= 0)
            if stop.size:
                count = int(stop[0])
        r = int(np.searchsorted(f["row_entry"], count))   # cuts of the first `count` entries = a prefix of the block
        nnz = int(f["indptr"][r])
        return f["indptr"][:r + 1], f["indices"][:nnz], f["values"][:nnz], f["rhs"][:r]

    # -- data access -----------------------------------------------------------------
    def _need(self, upto):
        """The head came from the device's top-k selection; anything beyond it asks for the
        complete ranking (full device sort), once."""
        upto = min(upto, self._n)
        if upto <= self._have:
            return
        if self._b.point_token != self._point:
            raise RuntimeError("rank list is stale: the device now holds the scores of a newer LP point")
        self._b.drain()
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

    def count_above(self, thr):
        """Number of entries of the WHOLE list whose score exceeds ``thr`` (cut_select_qcqp.py:85-87 counts the entries above
        BIG_M).  The list is sorted by score, so the head answers when its last score is <= thr; under the combined strategy a
        head that consists of marked entries only (obj_improve + BIG_M, cut_select_qp.py:611) answers too when the entries behind
        it -- the ones the scan never reached, all with an obj_improve no larger than the last marked one's -- cannot exceed thr.
        Otherwise the complete ranking is fetched."""
        h = self._score
        if self._have == 0:
            return 0
        last = float(h[self._have - 1])
        if self._have >= self._n or last <= thr or (self._strat == 4 and last > _BIG_M and last - _BIG_M <= thr):
            return int(np.count_nonzero(h[:self._have] > thr))
        return int(np.count_nonzero(self.scores() > thr))

    def _sets(self, lo, hi, loc):
        """(index sets [m, 5], sizes [m]) of positions [lo, hi): from the fused round's head when it covers them"""
        if self._head_sets is not None and hi <= self._head_sets.shape[0]:
            return self._head_sets[lo:hi], self._head_ks[lo:hi]
        return self._b.sets_of(loc)

    def _entry(self, pos):
        idx, score = int(self._idx[pos]), float(self._score[pos])
        if self._b.agg_list is None or isinstance(self._b.agg_list, DeviceAgg):
            S, ks = self._sets(pos, pos + 1, np.array([idx - self._b.scorer.base], dtype=np.int64))
            k, n = int(ks[0]), self._b.nb_vars
            set_inds = [int(v) for v in S[0, :k]]
            Xarr_inds = [n * set_inds[a] - set_inds[a] * (set_inds[a] + 1) // 2 + set_inds[c] for a in range(k) for c in range(a, k)]
        else:
            set_inds, Xarr_inds = self._b.agg_entry(idx)
        if self._kind == 1:
            return FeasEntry((set_inds, score, Xarr_inds, len(set_inds)), idx, self._b)
        L = self._b.nb_lifted
        curr_pt = tuple(self._vv[L + i] for i in set_inds)
        X_slice = tuple(self._vv[i] for i in Xarr_inds)
        return (idx, score, curr_pt, X_slice)

    def _entries(self, lo, hi):
        """Entry tuples of positions [lo, hi) built with array gathers (one fancy-indexing pass per
        candidate size instead of a Python loop per value): what iteration and slicing use."""
        hi = min(hi, self._n)
        if hi <= lo:
            return []
        if hi - lo < 8:
            return [self._entry(p) for p in range(lo, hi)]
        b, vv, L, n = self._b, self._vv, self._b.nb_lifted, self._b.nb_vars
        ids = np.asarray(self._idx[lo:hi], dtype=np.int64)
        loc = ids - b.scorer.base
        scores = np.asarray(self._score[lo:hi], dtype=np.float64).tolist()
        S_all, ks = self._sets(lo, hi, loc)
        out = [None] * (hi - lo)
        for k in np.unique(ks):
            k = int(k)
            m = np.nonzero(ks == k)[0]
            S = S_all[m, :k].astype(np.int64)
            ia, ib = np.triu_indices(k)
            pos = n * S[:, ia] - S[:, ia] * (S[:, ia] + 1) // 2 + S[:, ib]
            if self._kind == 1:
                for j, s_row, p_row in zip(m.tolist(), S.tolist(), pos.tolist()):
                    i = int(ids[j])
                    if b.agg_list is not None and not isinstance(b.agg_list, DeviceAgg):      # hand out the caller's own lists, as the reference does
                        s_row, p_row = b.agg_list[int(loc[j])][0:2]
                    out[j] = FeasEntry((s_row, scores[j], p_row, k), i, b)
            else:
                cp, Xs = vv[L + S].tolist(), vv[pos].tolist()
                for t, j in enumerate(m.tolist()):
                    out[j] = (int(ids[j]), scores[j], tuple(cp[t]), tuple(Xs[t]))
        return out

    # -- Sequence protocol -----------------------------------------------------------
    def __len__(self):
        return self._n

    def __getitem__(self, key):
        if isinstance(key, slice):
            rng = range(*key.indices(self._n))
            if not len(rng):
                return []
            self._need((rng[-1] if rng.step > 0 else rng[0]) + 1)      # (max(rng) walks the range: 16 us for a head of 5000)
            if rng.step == 1 and rng.start == 0:
                return RankListHead(self, rng.stop)
            if rng.step == 1:
                return self._entries(rng.start, rng.stop)
            return [self._entry(p) for p in rng]
        if key < 0:
            key += self._n
        if not 0 <= key < self._n:
            raise IndexError(key)
        self._need(key + 1)
        return self._entry(key)

    def __iter__(self):
        for lo in range(0, self._n, 65536):
            hi = min(lo + 65536, self._n)
            self._need(hi)
            for e in self._entries(lo, hi):
                yield e

    def __add__(self, other):
        """``a + b`` of cut_select_qcqp.py:79 without building either list: the caller slices
        ``[0:sel_size]`` off the concatenation, and only that slice is materialised."""
        return _Concat([self, other])

    def __radd__(self, other):
        return _Concat([other, self])


class RankListHead(Sequence):
    """``rank_list[0:count]`` (cut_select_qp.py:181 hands the whole list over, cut_select_qcqp.py:79-97 slices of
    it): the entries are built when somebody looks at them, and cut generation still finds the rows the fused
    round assembled for exactly these candidates."""

    def __init__(self, parent, count):
        self.parent, self._n = parent, int(count)
        self._list = None

    def _entries(self):
        if self._list is None:
            self._list = self.parent._entries(0, self._n)
        return self._list

    def __len__(self):
        return self._n

    def __getitem__(self, key):
        if isinstance(key, slice):
            start, stop, step = key.indices(self._n)
            if start == 0 and step == 1:
                return RankListHead(self.parent, stop)
        return self._entries()[key]

    def __iter__(self):
        return iter(self._entries())

    def __eq__(self, other):
        return list(self) == list(other)

    def __add__(self, other):
        return _Concat([self, other])

    def __radd__(self, other):
        return _Concat([other, self])


class _Concat(Sequence):
    """Lazy concatenation of rank lists (``rank_list_comb_obj + rank_list_feas_cons``)."""

    def __init__(self, parts):
        self._parts = [p for p in parts]

    def __len__(self):
        return sum(len(p) for p in self._parts)

    def __iter__(self):
        for p in self._parts:
            for e in p:
                yield e

    def __getitem__(self, key):
        n = len(self)
        if isinstance(key, slice):
            start, stop, step = key.indices(n)
            if step != 1:
                return [self[i] for i in range(start, stop, step)]
            # a slice stays a concatenation of (slices of) its parts: heads of rank lists keep the link to the cuts
            # their fused rounds assembled (cut_select_qcqp.py:79 slices (A + B)[0:sel_size] before generating)
            sub, off = [], 0
            for p in self._parts:
                lo, hi = max(start - off, 0), min(stop - off, len(p))
                if lo < hi:
                    sub.append(p[lo:hi])
                off += len(p)
            return _Concat(sub)
        if key < 0:
            key += n
        off = 0
        for p in self._parts:
            if key - off < len(p):
                return p[key - off]
            off += len(p)
        raise IndexError(key)

    def __eq__(self, other):
        return list(self) == list(other)

    def __add__(self, other):
        return _Concat(self._parts + [other])

    def __radd__(self, other):
        return _Concat([other] + self._parts)


class _Binding(object):
    """Device-side twin of one ``agg_list``: a Scorer handle holding its index sets."""

    def __init__(self, scorer, agg_list, nb_vars, nb_lifted):
        self.scorer, self.agg_list = scorer, agg_list
        self.nb_vars, self.nb_lifted = nb_vars, nb_lifted
        self.rank_serial = 0
        self.point_token = None  # identifies the LP point whose scores the device holds
        self.point_copy = None   # copy of the LP point last uploaded through this binding
        self.point_serial = 0
        self.scored = 0
        self.serial = 0
        self.set_arr = None      # [N, 5] index sets as arrays (vectorised entry building)
        # a fused round begun on the scorer and not yet ended (sdpcut_round_csr_begin): (strat, head, LP point)
        self._pending = None
        # (binding, head) ranked by feasibility right after this one at the same LP point last time (the QCQP round's
        # second cover, cut_select_qcqp.py:75-76): its round is begun together with this one's
        self.follower = None
        self.leader = None       # the binding this one follows (a follower never leads: no cycles)
        self.wasted = 0          # speculative rounds nobody asked for
        self.prioritised = False # its scorer's stream has been given the device's highest priority (a short list beside a long one)

    def begin(self, strat, head, vv, flags):
        token = self.scorer.round_csr_begin(strat, head, point=vv)
        self.note_point(vv, flags)
        self._pending = (token, strat, head, self.point_copy)

    @property
    def pending(self):
        """(strat, head, LP point) of the round begun through this binding and still pending on its scorer, or None"""
        p = self._pending
        if p is not None and self.scorer.pending is not p[0]:       # somebody ended or dropped it on the scorer
            p = self._pending = None
        return None if p is None else p[1:]

    def end(self):
        self._pending = None
        return self.scorer.round_csr_end()

    def drain(self):
        """end a round nobody collected (the scorer's block and scores are about to be used for something else)"""
        if self.pending is not None:
            self.end()
            self.wasted += 1
            if self.leader is not None:                   # the pair is not a pattern after all
                self.leader.follower, self.leader = None, None

    def note_point(self, vv, scored=0):
        """the device now holds LP point ``vv`` (and the measures ``scored`` at it)"""
        self.point_serial += 1
        self.point_token = self.point_serial
        self.point_copy = np.array(vv, dtype=np.float64)
        self.scored = scored

    def agg_entry(self, idx):
        if self.agg_list is not None:
            e = self.agg_list[idx]
            return e[0], e[1]
        S, ks = self.sets_of(np.array([idx], dtype=np.int64))
        k = int(ks[0])
        s = [int(v) for v in S[0, :k]]
        n = self.nb_vars
        pos = [n * s[a] - s[a] * (s[a] + 1) // 2 + s[b] for a in range(k) for b in range(a, k)]
        return s, pos

    def sets_of(self, local_ids):
        """(index sets [m, 5], sizes [m]) of candidates by local index: from the host arrays the
        list was bound from, or fetched from the device for lists that only exist there."""
        if self.set_arr is not None:
            return self.set_arr[local_ids], self.ks[local_ids]
        self.drain()
        return self.scorer.get_candidates(local_ids)


class AggArrays(Sequence):
    """Array-backed ``agg_list``: the candidate records of cut_select_qp.py:529-540 built on
    demand.  A Python list of 1.7e6 tuples (spar125-075-1, dim 4) takes minutes to build and
    GBs to hold; the loop only ever reads ``[0]`` / ``[1]`` of the few thousand selected entries."""

    def __init__(self, set_inds, ks, nb_vars, Q_arr=None):
        self.set_inds = np.ascontiguousarray(set_inds, dtype=np.int32)
        self.ks = np.ascontiguousarray(ks, dtype=np.int32)
        self.nb_vars, self.Q_arr = nb_vars, Q_arr
        self.serial = 0          # bumped by shuffle(): device bindings of the old order are stale

    def __len__(self):
        return self.ks.shape[0]

    def shuffle(self):
        """In-place random reordering, the array form of ``np.random.shuffle(agg_list)``
        (cut_select_qp.py:636): ``np.random.permutation(N)`` draws the same permutation from the
        global legacy generator as the list shuffle would (same seed -> same order)."""
        perm = np.random.permutation(len(self))
        self.set_inds = np.ascontiguousarray(self.set_inds[perm])
        self.ks = np.ascontiguousarray(self.ks[perm])
        self.serial += 1

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


class DeviceAgg(Sequence):
    """``agg_list`` whose index sets were enumerated on the device and never came to the host
    (``sdpcut_set_candidates_cover``): a length, and records fetched on demand for the few entries
    somebody indexes.  Bound to the scorer that holds the list."""

    def __init__(self, scorer, count, nb_vars, Q_arr=None):
        self.scorer, self._n, self.nb_vars, self.Q_arr = scorer, int(count), nb_vars, Q_arr
        self.serial = 0

    def __len__(self):
        return self._n

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            rng = range(*idx.indices(self._n))
            S, ks = self.scorer.get_candidates(np.fromiter(rng, dtype=np.int64, count=len(rng)))
            return [AggArrays(S[i:i + 1], ks[i:i + 1], self.nb_vars, self.Q_arr)[0] for i in range(len(rng))]
        if idx < 0:
            idx += self._n
        if not 0 <= idx < self._n:
            raise IndexError(idx)
        S, ks = self.scorer.get_candidates(np.array([idx], dtype=np.int64))
        return AggArrays(S, ks, self.nb_vars, self.Q_arr)[0]

    def to_arrays(self):
        """the whole list on the host (tests; the random strategy reorders a host copy)"""
        S, ks = self.scorer.get_candidates(np.arange(self._n, dtype=np.int64))
        return AggArrays(S, ks, self.nb_vars, self.Q_arr)


class GpuCutSelectionMixin(object):
    """The four hot-path methods of the reference's ``CutSolver`` on the GPU."""

    _gpu_device = 0
    _sparse_pair = None
    _gpu_overlap = True      # begin the follower list's round together with the leader's (QCQP: two covers per LP point)

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

    def _gpu_wake(self):
        """Poke the device(s) of the bound candidate lists (sdpcut_wake): called the moment an LP solve returns, while the solution
        vector is still being extracted -- a round issued to a device that idled through the solve costs 0.1-0.2 ms more than one
        issued back to back, and this gets 40-65 us of it back (bench.py: secondary.cold_round).  A maintainer of the reference's
        own loop adds `self._gpu_wake()` behind `my_prob.solve()` (cut_select_qp.py:194); optional, changes no result."""
        for b in getattr(self, "_gpu_bindings", {}).values():
            b.scorer.wake()

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
        if b is not None and b.agg_list is agg and b.n_at_bind == len(agg) and b.serial == getattr(agg, "serial", 0):
            return b
        N = len(agg)
        if isinstance(agg, DeviceAgg):          # the list already lives on its scorer
            b = _Binding(agg.scorer, agg, self._nb_vars, self._nb_lifted)
            b.n_at_bind, b.serial = N, agg.serial
            self._gpu_bindings[key] = b
            return b
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
        b.ks, b.set_arr = ks[:N], S[:N]
        b.serial = getattr(agg, "serial", 0)
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
    def _gpu_point(b, vars_values, flags, cut_round=None):
        """Make ``vars_values`` the device's LP point and score what ``flags`` asks for.

        The upload is skipped only when the point equals -- entry by entry -- the copy kept of the last one
        uploaded through this binding (an in-place edit of the caller's array is a new point); the upload
        itself is an asynchronous copy out of pinned staging, so a redundant one costs little.
        :meth:`invalidate_point` forces the next call to upload."""
        b.drain()
        vv = vars_values if (isinstance(vars_values, np.ndarray) and vars_values.dtype == np.float64 and
                             vars_values.flags.c_contiguous) else np.ascontiguousarray(vars_values, dtype=np.float64)
        same = b.point_copy is not None and b.point_copy.shape == vv.shape and np.array_equal(vv, b.point_copy)
        if not same:
            b.scorer.set_point(vv)
            b.note_point(vv)
        need = flags & ~b.scored
        if need:
            b.scorer.score(need)
            b.scored |= need
        return vv

    def invalidate_point(self):
        """Forget which LP point the device holds: the next selection / generation uploads its point whatever it is
        (for callers that change device state behind the mixin's back, e.g. through the Scorer directly)."""
        for b in getattr(self, "_gpu_bindings", {}).values():
            b.drain()
            b.point_copy, b.scored = None, 0

    # ------------------------------------------------------------------ a7-a9 selection
    def _sel_eigcut_by_ordering_on_measure(self, strat, vars_values, cut_round, sel_size=0):
        """Strategies 1 (feasibility), 2 (optimality via MLP), 4 (combined), 5 (random);
        same returns as cut_select_qp.py:543-703.  Strategies 3 / -1 need an exact SDP
        solver per candidate and are out of scope."""
        if strat == 5:
            # random order, in place, like :634-637; the device twin of the old order is dropped
            agg = self._agg_list
            if isinstance(agg, DeviceAgg):
                agg = self._agg_list = agg.to_arrays()
            if isinstance(agg, AggArrays):
                agg.shuffle()
            else:
                np.random.shuffle(agg)
            getattr(self, "_gpu_bindings", {}).pop(id(agg), None)
            return agg
        if strat not in (1, 2, 4):
            raise NotImplementedError("exact-SDP strategies (3, -1) are not part of the GPU path")
        b = self._gpu_bind()
        N = len(self._agg_list)
        sel_size = min(sel_size, N)
        flags = {1: _capi.EIG, 2: _capi.NN, 4: _capi.EIG | _capi.NN}[strat]
        if N == 0:
            return []       # strat 4 included: sel_size is clamped to 0 and the reference falls through
        # head fetched eagerly: what the loop can consume (sel_size is only passed for strat 4;
        # for 1 / 2 the cap of :37 bounds it)
        head = min(N, sel_size if (strat == 4 and sel_size > 0) else _HEAD)
        vv = vars_values if (isinstance(vars_values, np.ndarray) and vars_values.dtype == np.float64 and
                             vars_values.flags.c_contiguous) else np.ascontiguousarray(vars_values, dtype=np.float64)
        fused = None
        # which list is ranked by feasibility right after which at the same point: the QCQP loop's pair (cut_select_qcqp.py:64-76)
        last = getattr(self, "_gpu_last", None)
        if (self._gpu_overlap and strat == 1 and last is not None and last[0] is not b and last[0].follower is None
                and last[0].leader is None and b.follower is None and b.leader is None and 1 <= head <= _FUSED_HEAD_MAX and last[1].shape == vv.shape and np.array_equal(last[1], vv)):
            last[0].follower, b.leader = (b, head), last[0]
        if 1 <= head <= _FUSED_HEAD_MAX and not (strat == 4 and sel_size == 0):
            # ONE library call for the whole separation step (cut_select_qp.py:165-182): LP point up, scores, ranking
            # AND the assembled cuts of the head, which _gen_eigcuts_selected then only hands to the LP.  In two halves:
            # between them the follower list's round is begun too, so both covers' device work overlaps.
            pend = b.pending
            if pend is not None and not (pend[0] == strat and pend[1] == head and pend[2].shape == vv.shape and np.array_equal(pend[2], vv)):
                b.drain()
                pend = None
            f = b.follower
            spec = f is not None and f[0].pending is None and f[0] is not b
            if spec and b.n_at_bind < f[0].n_at_bind:
                # the longer list's round is begun first; the shorter one's few small kernels go ahead of its waiting
                # workgroups on a high-priority stream (SDPCUT_OPT_STREAM_PRIORITY), so this call still returns early
                if not b.prioritised:
                    b.drain()
                    b.scorer.set_option(_capi.OPT_STREAM_PRIORITY, 1)
                    b.prioritised, pend = True, None
                f[0].begin(1, f[1], vv, _capi.EIG)
                spec = False
            if pend is None:
                b.begin(strat, head, vv, flags)
            if spec:
                if f[0].n_at_bind < b.n_at_bind and not f[0].prioritised:      # (the other way round: the follower is the short list)
                    f[0].scorer.set_option(_capi.OPT_STREAM_PRIORITY, 1)
                    f[0].prioritised = True
                f[0].begin(1, f[1], vv, _capi.EIG)
            fused = b.end()
            idx, score = fused["idx"].copy(), fused["score"].copy()
            total, new_strat, counters = fused["n_total"], fused["new_strat"], fused["counters"]
            self._gpu_last = (b, b.point_copy)
        else:
            self._gpu_last = None
            vv = self._gpu_point(b, vv, flags, cut_round)
            idx, score, total, new_strat, counters = b.scorer.rank(strat, sel_size, head)
        b.rank_serial += 1
        rl = RankList(self, b, 1 if strat == 1 else 2, total, vv, idx, score, strat=strat, sel_size=sel_size, fused=fused)
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
        if sel_size <= 0:
            self._my_prob.linear_constraints.add(lin_expr=[], rhs=[], senses=[])
            return 0
        # (binding, candidate indices, point) groups in list order; most lists have one group
        groups, vv = None, vars_values
        if isinstance(rank_list, _Concat):
            # (A + B)[0:sel_size] of the QCQP round: the cuts of each part, in list order, if every part still has them
            blocks, left = [], sel_size
            for part in rank_list._parts:
                if left <= 0:
                    break
                rl = part.parent if isinstance(part, RankListHead) else part
                csr = None
                if isinstance(rl, RankList):
                    csr = rl.fused_rows(min(left, len(part)), strong_only=opt_sel and strong_only and rl._kind != 1,
                                        vars_values=vars_values if rl._kind == 1 else None)
                if csr is None:
                    blocks = None
                    break
                blocks.append(csr)
                left -= min(left, len(part))
            if blocks is not None:
                return sum(self._gpu_add_csr(c, pair) for c in blocks)
        if isinstance(rank_list, RankListHead):
            rank_list, sel_size = rank_list.parent, min(sel_size, len(rank_list))
        if isinstance(rank_list, RankList):
            csr = rank_list.fused_rows(sel_size, strong_only=opt_sel and strong_only,
                                       vars_values=vars_values if feas_sel else None)
            if csr is not None:
                return self._gpu_add_csr(csr, pair)
            idx, vv = rank_list.ids(sel_size), rank_list._vv
            if opt_sel and strong_only:                       # :725-726
                stop = np.nonzero(rank_list.scores(sel_size) <= 0)[0]
                if stop.size:
                    idx = idx[:stop[0]]
            groups = [(rank_list._b, idx)]
        elif strat == 5 and rank_list is self._agg_list:
            # random selection (:729-732): the shuffled list itself, entries 0 .. sel_size-1
            groups = [(self._gpu_bind(), np.arange(sel_size, dtype=np.int64))]
        else:
            entries = list(rank_list[0:sel_size])
            if opt_sel:
                if strong_only:
                    cut = next((p for p, e in enumerate(entries) if e[1] <= 0), len(entries))
                    entries = entries[:cut]
                groups = [(self._gpu_bind(), np.array([e[0] for e in entries], dtype=np.int64))]
            elif feas_sel and all(isinstance(e, FeasEntry) and e.binding is not None for e in entries):
                # runs of entries of one candidate list (the QCQP feasibility round concatenates the
                # lists of two covers, cut_select_qcqp.py:79)
                groups = []
                for e in entries:
                    if groups and groups[-1][0] is e.binding:
                        groups[-1][1].append(e.agg_idx)
                    else:
                        groups.append((e.binding, [e.agg_idx]))
                groups = [(g, np.array(ix, dtype=np.int64)) for g, ix in groups]
            if groups is None:
                return self._gen_from_entries(entries, feas_sel, vars_values, pair)
        if not opt_sel or vv is None:
            vv = vars_values
        parts = []
        for g, idx in groups:
            if idx.size:
                self._gpu_point(g, vv, 0)
                parts.append(g.scorer.cut_rows(idx - g.scorer.base))
        if not parts:
            self._my_prob.linear_constraints.add(lin_expr=[], rhs=[], senses=[])
            return 0
        lam, coef, rhs, cols, ks = (np.concatenate(a) for a in zip(*parts)) if len(parts) > 1 else parts[0]
        keep = np.nonzero(lam < _THRES_NEG_EIGVAL)[0]          # :743
        store = self._my_prob.linear_constraints
        if hasattr(store, "add_csr"):
            # batched assembly (SURVEY 8 f row 4): the padded device rows become one CSR block with
            # array operations, no Python object per cut
            indptr, ind, val = rows_to_csr(coef[keep], cols[keep], ks[keep])
            store.add_csr(indptr, ind, val, rhs[keep], "G")
            return int(keep.size)
        rows, rhs_out = [], []
        for c in keep:
            w = int(ks[c]) * (int(ks[c]) + 3) // 2
            rows.append(pair(ind=cols[c, :w].tolist(), val=coef[c, :w].tolist()))
            rhs_out.append(float(rhs[c]))
        store.add(lin_expr=rows, rhs=rhs_out, senses=["G"] * len(rows))
        return len(rows)

    def _gpu_add_csr(self, csr, pair):
        """Hand a block of assembled cuts to the LP: as arrays when the row store takes them (``add_csr``), else as the
        reference's per-row objects (cut_select_qp.py:747-754).  The arrays are copied: they are views of the scorer's
        pinned block, which the next round overwrites."""
        indptr, ind, val, rhs = csr
        r = rhs.shape[0]
        store = self._my_prob.linear_constraints
        if hasattr(store, "add_csr"):
            store.add_csr(np.array(indptr), np.array(ind), np.array(val), np.array(rhs), "G")      # (int32 index arrays as the device wrote them)
            return r
        ptr, ind, val = indptr.tolist(), ind.tolist(), val.tolist()
        rows = [pair(ind=ind[ptr[c]:ptr[c + 1]], val=val[ptr[c]:ptr[c + 1]]) for c in range(r)]
        store.add(lin_expr=rows, rhs=rhs.tolist(), senses=["G"] * r)
        return r

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
        self._gpu_tri_triples64 = None
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
        if nb_tri_cuts <= 0:
            self._my_prob.linear_constraints.add(lin_expr=[], rhs=[], senses=[])
            return 0
        # the rows as ONE CSR block built with array operations (the per-row Python loop of :847-861 cost ~3 us per cut, 30 ms for
        # the 10 000 cuts of a round -- two orders of magnitude more than the device took to find them): inequality c of triple
        # (a, b, d) has columns X_ab, X_ad, X_bd and x_(a|b|d)[c]  (c < 3: 4 non-zeros, rhs 0) or x_a, x_b, x_d (c = 3: 6, rhs -1)
        e = np.asarray(ent[:nb_tri_cuts], dtype=np.int64)
        c = (e & 3).astype(np.int64)
        tri64 = getattr(self, "_gpu_tri_triples64", None)
        if tri64 is None or tri64.shape[0] != len(tri):
            tri64 = self._gpu_tri_triples64 = np.asarray(tri, dtype=np.int64)      # (converted once: 80 us per round otherwise)
        abd = tri64[e >> 2]
        a3, b3, d3 = abd[:, 0], abd[:, 1], abd[:, 2]
        ra, rb = n * a3 - a3 * (a3 + 1) // 2, n * b3 - b3 * (b3 + 1) // 2
        four = c < 3
        indptr = np.zeros(e.shape[0] + 1, dtype=np.int64)
        np.cumsum(np.where(four, 4, 6), out=indptr[1:])
        pos = indptr[:-1]
        ind = np.empty(int(indptr[-1]), dtype=np.int64)
        val = np.empty(int(indptr[-1]), dtype=np.float64)
        ind[pos], ind[pos + 1], ind[pos + 2] = ra + b3, ra + d3, rb + d3            # Xarr_inds[1], [2], [4]
        p4, p6 = pos[four], pos[~four]
        ind[p4 + 3] = abd[four, c[four]] + L
        ind[p6 + 3], ind[p6 + 4], ind[p6 + 5] = a3[~four] + L, b3[~four] + L, d3[~four] + L
        v4 = np.array([[-1.0, -1.0, 1.0, 1.0], [-1.0, 1.0, -1.0, 1.0], [1.0, -1.0, -1.0, 1.0]])[c[four]]
        for j in range(4):
            val[p4 + j] = v4[:, j]
        for j, v in enumerate((1.0, 1.0, 1.0, -1.0, -1.0, -1.0)):
            val[p6 + j] = v
        rhs = np.where(four, 0.0, -1.0)
        store = self._my_prob.linear_constraints
        if hasattr(store, "add_csr"):
            store.add_csr(indptr, ind, val, rhs, "G")
            return int(e.shape[0])
        ptr, ind_l = indptr.tolist(), ind.tolist()
        val_l = [int(v) for v in val.tolist()]                    # the reference hands integer coefficients (:833-836)
        rows = [pair(ind=ind_l[ptr[r]:ptr[r + 1]], val=val_l[ptr[r]:ptr[r + 1]]) for r in range(e.shape[0])]
        store.add(lin_expr=rows, rhs=[int(v) for v in rhs.tolist()], senses=["G"] * len(rows))
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

    # ------------------------------------------------------------------ rounds without CPLEX (SURVEY 8 f row 2)
    _CONVERGENCE_TOL = 10 ** (-3)         # cut_select_qp.py:29

    def cut_select_algo(self, filename, dim, sel_size, strat=2, nb_rounds_cuts=20, term_on=False,
                        triangle_on=False, strong_only=False, max_subs=_THRES_MAX_SUBS, on_round=None):
        """Cutting-plane rounds on a BoxQP ``.in`` file, same arguments and default return tuple as
        the reference's entry point (cut_select_qp.py:73-221), with HiGHS as LP solver, the native
        cover enumeration and the GPU selection / generation / triangle separation in between.
        ``max_subs=None`` lifts the reference's 4e6 candidate guard (:117-120); ``on_round(r, log)`` is
        called after every LP solve (progress of long runs).  Dense cuts
        (strat 0), exact-SDP strategies and chordal extensions are out of scope.
        -> (bound per solve, total s, round s, separation s, PSD cuts per round, triangle cuts per
        round, number of candidates)."""
        from timeit import default_timer as clock
        from . import harness
        if strat not in (1, 2, 4, 5):
            raise AssertionError("strategies on the GPU path: 1 feasibility, 2 optimality, 4 combined, 5 random")
        assert 0 < sel_size, "The selection size must be a % or number (of cuts) >0!"
        assert dim <= 5, "Keep SDP vertex cover low-dimensional (<=5)!"
        t_start = clock()
        inst = harness.parse_boxqp(filename)
        self._dim = dim
        self._nb_vars, self._nb_lifted, self._Q_arr, self._Q_adj = (inst[k] for k in ("nb_vars", "nb_lifted", "Q_arr", "adj"))
        self._gpu_nets = {}
        if strat in (2, 4):
            self._load_neural_nets()
        # the cover is enumerated on the device, straight into the scorer's candidate list
        sc = self._gpu_new_scorer()
        sc.set_instance(self._nb_vars, np.asarray(self._Q_arr, dtype=np.float64))
        n_cand = sc.set_candidates_cover(inst["adj"], dim, max_subs=max_subs or 0)
        if (max_subs and n_cand >= max_subs) or nb_rounds_cuts == 0:
            sc.close()
            return [0, 0], clock() - t_start, 0, 0, [0], 0, n_cand                      # the reference's guard tuple
        self._agg_list = DeviceAgg(sc, n_cand, self._nb_vars, self._Q_arr)
        quota = self.selection_size(sel_size, n_cand)
        t_model = clock()
        self._my_prob = lp = harness.boxqp_relaxation(inst)
        t_model = clock() - t_model
        if triangle_on:
            self._preprocess_triangle_ineq()
        state = {"strat": strat}

        def separate(round_no, point):
            cur = state["strat"]
            picked = self._sel_eigcut_by_ordering_on_measure(cur, point, round_no, **({"sel_size": quota} if cur == 4 else {}))
            if cur == 4 and isinstance(picked, tuple):
                state["strat"], picked = picked       # the switch takes effect next round (:181 vs :188)
            sdp = self._gen_eigcuts_selected(cur, quota, picked, strong_only=strong_only, vars_values=point)
            tri = self._separate_and_add_triangle(sel_size, point) if triangle_on else 0
            return {"sdp": sdp, "tri": tri}

        log = harness.run_cut_rounds(lp, separate, nb_rounds_cuts, setup_s=t_model,
                                     stop_tol=self._CONVERGENCE_TOL if term_on else None, on_round=on_round,
                                     after_solve=self._gpu_wake)
        sep = [t_model] + log.separation_s
        return ([-v for v in log.bounds], clock() - t_start, [a + b for a, b in zip(log.solve_s, [0.0] + log.separation_s)],
                sep, [0] + log.column("sdp"), log.column("tri"), n_cand)


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
        nb_opt_cuts = (comb_obj.count_above(_BIG_M) if isinstance(comb_obj, RankList) else
                       int(np.count_nonzero(np.array([e[1] for e in comb_obj]) > _BIG_M))) if len(comb_obj) else 0
        nb_cuts_combined = n_obj      # :90-92 counts the entries whose first field is an int: the objective cover's (optimality / combined entries)
        rest = sel_size - nb_cuts_combined
        nb_a = self._gen_eigcuts_selected(1, rest, feas_cons[0:rest], vars_values=vars_values)
        nb_b = self._gen_eigcuts_selected(strat_old, nb_cuts_combined, comb_obj[0:nb_cuts_combined],
                                          vars_values=vars_values)
        return strat, rank_list, nb_a + nb_b, nb_opt_cuts

    def cut_select_algo(self, filename, dim, sel_size=0.1, strat=2, nb_rounds_cuts=20):
        """Cutting-plane rounds on a QCQP in OSiL format, same arguments and return tuple as the
        reference's QCQP entry point (cut_select_qcqp.py:16-113), with HiGHS, the native enumeration
        of both covers and :meth:`select_and_generate_round` between two solves.
        -> (objective value per solve, sel_size, PSD cuts per round, optimality cuts per round)."""
        from . import harness
        if strat not in (1, 2, 4, 5):
            raise AssertionError("strategies on the GPU path: 1 feasibility, 2 optimality, 4 combined, 5 random")
        assert 0 < sel_size, "The selection size must be a % or number (of cuts) >0!"
        assert dim <= 5, "Keep SDP vertex cover low-dimensional (<=5)!"
        inst = harness.parse_osil(filename)
        self._dim = dim
        self._nb_vars, self._nb_lifted, self._Q_arr = inst["nb_vars"], inst["nb_lifted"], inst["Q_arr"]
        self._Q_adj, self._Q_adj_cons = inst["adj"], inst["adj_cons"]
        self._my_prob = lp = harness.LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
        lp.linear_constraints.add(inst["rows"], inst["rhs"], inst["senses"])
        lp.linear_constraints.add_csr(*harness.mccormick_csr(self._nb_vars, inst["adj"]), "L")
        self._load_neural_nets()
        # both covers and their intersection on the device (:50, :314-334): the lists never come to the host
        sc_o, sc_c = self._gpu_new_scorer(), self._gpu_new_scorer()
        for sc in (sc_o, sc_c):
            sc.set_instance(self._nb_vars, np.asarray(self._Q_arr, dtype=np.float64))
        n_o, n_c = sc_o.set_candidates_cover_split(sc_c, inst["adj"], inst["adj_cons"], dim)
        cover_obj = DeviceAgg(sc_o, n_o, self._nb_vars, self._Q_arr)
        cover_cons = DeviceAgg(sc_c, n_c, self._nb_vars, self._Q_arr)
        if strat == 5:
            cover_obj = cover_obj.to_arrays()       # the random strategy reorders its list in place, round after round (:634-637)
        self._agg_list = cover_obj
        quota = self.selection_size(sel_size, len(cover_obj), minimum=1)                         # :55-58
        state = {"strat": strat}

        def separate(round_no, point):
            state["strat"], _, sdp, opt = self.select_and_generate_round(state["strat"], point, round_no, quota,
                                                                         cover_obj, cover_cons)
            return {"sdp": sdp, "opt": opt}

        log = harness.run_cut_rounds(lp, separate, nb_rounds_cuts, after_solve=self._gpu_wake)
        opt = log.column("opt")
        return log.bounds, quota, [0] + log.column("sdp"), [0] + opt + [0] * (nb_rounds_cuts - len(opt))


def make_dropin_classes(cut_select_qp, cut_select_qcqp=None):
    """Compose the GPU mixin with the reference's own classes (modules passed in, nothing is
    imported here) -> (GpuCutSolver, GpuCutSolverQCQP or None).

    The QCQP loop reaches the hot path through ``super()`` from inside ``CutSolverQCQP``
    (cut_select_qcqp.py:41, :66-76), i.e. through whatever FOLLOWS ``CutSolverQCQP`` in the
    instance's MRO.  Putting the mixin in front of ``CutSolverQCQP`` would leave those calls on
    the reference's CPU loop; the mixin has to sit between the two reference classes:

        GpuCutSolver     = (GpuCutSelectionMixin, CutSolver)
        GpuCutSolverQCQP = (CutSolverQCQP, GpuCutSolver)
        MRO: GpuCutSolverQCQP, CutSolverQCQP, GpuCutSolver, GpuCutSelectionMixin, CutSolver, object
    """
    qp = type("GpuCutSolver", (GpuCutSelectionMixin, cut_select_qp.CutSolver), {"__doc__": "CutSolver with the hot path on the GPU"})
    qcqp = None
    if cut_select_qcqp is not None:
        qcqp = type("GpuCutSolverQCQP", (cut_select_qcqp.CutSolverQCQP, qp), {"__doc__": "CutSolverQCQP with the hot path on the GPU"})
    return qp, qcqp

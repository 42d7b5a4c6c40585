"""ctypes binding of libsdpcut_hip.so (C-ABI declared in include/sdpcut.h).

This is the only place where the host code touches native code, mirroring how the
reference reaches NNs.so through ctypes (cut_select_qp.py:284-303).  There is no CPU
fallback: a missing library or a missing gfx950 device raises.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDPCUT_LIB") or os.path.join(HERE, "libsdpcut_hip.so")   # env: experiment builds
# the reference's own FFI (six symbols of NNs.so) lives in a sibling library that binds to the one above privately
NNS_LIB_PATH = os.path.join(HERE, "libsdpcut_nns.so")

EIG, NN = 1, 2
STRAT_FEAS, STRAT_OPT, STRAT_COMB = 1, 2, 4
PART_STRONG = 104
PART_COMBALL = 105
KERNEL_MFMA, KERNEL_SIMPLE, KERNEL_VALU = 0, 1, 2
OPT_KERNEL, OPT_TIMING, OPT_FUSE_KEYS, OPT_AUTO_REGIME, OPT_FUSED_TAIL, OPT_COOP_LAUNCH, OPT_EIG_KERNEL, OPT_STREAM_PRIORITY, OPT_SIDE_STREAMS, OPT_ONE_LAUNCH, OPT_PREFILTER = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
STAT_ROUNDS, STAT_SELECT_FALLBACKS, STAT_SCORED, STAT_TIE_SPLITS, STAT_DIRECT_SELECTIONS, STAT_PF_BIN, STAT_PF_FLOOR, STAT_PF_COUNT = 1, 2, 3, 4, 5, 6, 7, 8
ROW_LD = 20

_c = ctypes
_i32p = _c.POINTER(_c.c_int32)
_i64p = _c.POINTER(_c.c_int64)
_dp = _c.POINTER(_c.c_double)
_vp = _c.c_void_p

class RoundCsr(_c.Structure):
    """sdpcut_round_csr_t of include/sdpcut.h"""
    _fields_ = [("cap", _c.c_int64), ("n_out", _c.c_int64), ("n_total", _c.c_int64), ("new_strat", _c.c_int32), ("row_ld", _c.c_int32),
                ("counters", _c.c_int64 * 4), ("idx", _vp), ("score", _vp), ("lam_min", _vp), ("ks", _vp), ("set_inds", _vp),
                ("n_rows", _c.c_int64), ("nnz", _c.c_int64), ("row_entry", _vp), ("indptr", _vp), ("indices", _vp),
                ("values", _vp), ("rhs", _vp)]


# name -> argtypes (restype is c_int unless listed in _RESTYPES); kept in one table so that the
# CPU test-suite can check that the library exports every symbol the header declares
SIGNATURES = {
    "sdpcut_version": [],
    "sdpcut_last_error": [_vp],
    "sdpcut_create": [_c.c_int, _c.POINTER(_vp)],
    "sdpcut_destroy": [_vp],
    "sdpcut_set_option": [_vp, _c.c_int, _c.c_int64],
    "sdpcut_set_stream": [_vp, _vp],
    "sdpcut_get_stat": [_vp, _c.c_int, _i64p],
    "sdpcut_synchronize": [_vp],
    "sdpcut_wake": [_vp],
    "sdpcut_set_network": [_vp, _c.c_int, _c.c_int, _i32p, _dp, _c.c_int64],
    "sdpcut_set_instance": [_vp, _c.c_int32, _dp],
    "sdpcut_set_candidates": [_vp, _c.c_int64, _i32p, _c.c_int32, _i32p, _c.c_int64],
    "sdpcut_set_candidates_philox": [_vp, _c.c_int32, _c.c_int64, _c.c_uint64, _c.c_int64],
    "sdpcut_set_candidates_cover": [_vp, _c.POINTER(_c.c_uint8), _c.c_int32, _c.c_int64, _i64p],
    "sdpcut_set_candidates_cover_split": [_vp, _vp, _c.POINTER(_c.c_uint8), _c.POINTER(_c.c_uint8), _c.c_int32, _i64p, _i64p],
    "sdpcut_get_candidates": [_vp, _c.c_int64, _i64p, _i32p, _i32p],
    "sdpcut_set_builtin_networks": [_vp, _c.c_int],
    "sdpcut_set_point": [_vp, _dp],
    "sdpcut_point_buffer": [_vp, _c.POINTER(_dp)],
    "sdpcut_set_point_device": [_vp, _vp],
    "sdpcut_score": [_vp, _c.c_uint32],
    "sdpcut_get_scores": [_vp, _dp, _dp],
    "sdpcut_rank": [_vp, _c.c_int, _c.c_int64, _c.c_int64, _i64p, _dp, _i64p, _i32p, _i64p],
    "sdpcut_rank_device": [_vp, _c.c_int, _c.c_int64, _c.c_int64, _vp, _vp, _i64p, _i64p, _i32p, _i64p],
    "sdpcut_rank_fetch": [_vp, _c.c_int64, _c.c_int64, _i64p, _dp],
    "sdpcut_merge_topk_device": [_vp, _c.c_int64, _vp, _vp, _vp, _c.c_int64, _vp, _vp],
    "sdpcut_gather_scores_device": [_vp, _c.c_int64, _vp, _vp, _vp],
    "sdpcut_cut_rows": [_vp, _c.c_int64, _i64p, _dp, _dp, _dp, _i64p, _i32p],
    "sdpcut_select_round": [_vp, _c.c_int, _c.c_int64, _c.c_int32, _i64p, _dp, _dp, _dp, _dp, _i32p, _i64p, _i64p, _i32p, _i64p],
    "sdpcut_select_round_view": [_vp, _c.c_int, _c.c_int64, _c.c_int32, _c.POINTER(_c.c_void_p), _i64p, _i64p, _i64p, _i32p, _i64p],
    "sdpcut_round_view": [_vp, _dp, _c.c_int, _c.c_int64, _c.c_int32, _c.POINTER(_c.c_void_p), _i64p, _i64p, _i64p, _i32p, _i64p],
    "sdpcut_round_csr": [_vp, _dp, _c.c_int, _c.c_int64, _c.POINTER(RoundCsr)],
    "sdpcut_round_csr_begin": [_vp, _dp, _c.c_int, _c.c_int64],
    "sdpcut_round_csr_end": [_vp, _c.POINTER(RoundCsr)],
    "sdpcut_shard_head_device": [_vp, _c.c_int, _c.c_int64, _vp],
    "sdpcut_shard_finish_enqueue": [_vp, _c.c_int32, _c.c_int64, _c.c_int32, _vp, _c.c_int64, _c.c_int64, _c.c_int32],
    "sdpcut_shard_finish_wait": [_vp, _c.c_int32, _c.POINTER(_c.c_void_p), _i64p],
    "sdpcut_shard_finish_round": [_vp, _c.c_int32, _c.c_int64, _vp, _c.c_int64, _c.c_int32, _i64p, _i64p, _dp, _dp, _dp, _dp, _i32p],
    "sdpcut_shard_finish_round_view": [_vp, _c.c_int32, _c.c_int64, _vp, _c.c_int64, _c.c_int32, _c.POINTER(_c.c_void_p)],
    "sdpcut_shard_finish_round_own": [_vp, _c.c_int32, _c.c_int64, _vp, _c.c_int64, _c.c_int32, _c.POINTER(_c.c_void_p), _i64p],
    "sdpcut_eig_batch": [_vp, _c.c_int, _c.c_int64, _dp, _dp, _dp, _dp],
    "sdpcut_nn_batch": [_vp, _c.c_int, _c.c_int64, _dp, _dp],
    "sdpcut_last_timing": [_vp, _dp, _c.c_int],
    "sdpcut_mfma_probe": [_vp, _dp, _dp, _dp],
    "sdpcut_tri_preprocess": [_vp, _c.POINTER(_c.c_uint8), _i64p],
    "sdpcut_tri_get_triples": [_vp, _i32p, _c.POINTER(_c.c_uint8)],
    "sdpcut_tri_separate": [_vp, _c.c_int64, _i64p, _dp, _i64p, _i64p],
    "sdpcut_enumerate_cover": [_c.c_int32, _c.POINTER(_c.c_uint8), _c.c_int32, _c.c_int64, _i32p, _i32p, _i64p],
}
_RESTYPES = {"sdpcut_last_error": _c.c_char_p}
# the reference's own FFI (cut_select_qp.py:297-303): ALL that libsdpcut_nns.so exports (include/sdpcut_nns.h)
COMPAT_SYMBOLS = ["neural_net_2D", "neural_net_3D", "neural_net_4D", "neural_net_5D", "NNs_initialize", "NNs_terminate"]

_lib = None


def _torch_installed():
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return sys.modules["torch"] is not None
    try:
        return importlib.util.find_spec("torch") is not None
    except (ImportError, ValueError):
        return False


def load_library(path=None):
    """dlopen the HIP library; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # PyTorch-ROCm bundles its own HIP/HSA runtime (same SONAME libamdhip64.so.7).  A process must hold exactly
    # one of them: where torch is installed it is imported first, so that this library binds to the runtime torch
    # (device memory, streams, torch.distributed/RCCL for the sharded rounds) brings in.  Where it is not -- the
    # reference itself needs numpy and ctypes only (cut_select_qp.py:1-14, requirements.txt) -- the library binds to the
    # system ROCm named in its RUNPATH: the single-GPU product has no torch dependency; `distributed.py` is the one
    # module that imports it.
    if _torch_installed():
        import torch  # noqa: F401
    if not os.path.exists(path):
        raise RuntimeError(
            "%s not found: build it with `python -m sdpcutsel_via_nn_amd.build` "
            "(there is no CPU fallback for the cut-scoring path)" % path)
    lib = ctypes.CDLL(path)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, _c.c_int)
    _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


class SdpCutError(RuntimeError):
    pass


def enumerate_cover(adjacency, dim, max_subs=None):
    """Index sets of the semidefinite vertex cover P^E_dim in the reference's order
    (cut_select_qp.py:399-524).  adjacency: [n, n] array, non-zero = edge.
    -> (set_inds int32 [N, 5] padded with -1, ks int32 [N], N).  If max_subs is given and
    N >= max_subs only the count is returned (arrays None), like the reference's RAM guard."""
    lib = load_library()
    adj = np.ascontiguousarray(np.asarray(adjacency) != 0, dtype=np.uint8)
    n = adj.shape[0]
    if adj.shape != (n, n):
        raise ValueError("adjacency must be square")
    cnt = _c.c_int64(0)
    u8 = adj.ctypes.data_as(_c.POINTER(_c.c_uint8))
    rc = lib.sdpcut_enumerate_cover(n, u8, int(dim), 0, None, None, ctypes.byref(cnt))
    if rc != 0:
        raise ValueError("sdpcut_enumerate_cover: bad arguments (dim must be 3..5)")
    N = cnt.value
    if max_subs is not None and N >= max_subs:
        return None, None, N
    sets = np.empty((max(N, 1), 5), dtype=np.int32)
    ks = np.empty(max(N, 1), dtype=np.int32)
    rc = lib.sdpcut_enumerate_cover(n, u8, int(dim), N, _ptr(sets, _i32p), _ptr(ks, _i32p), ctypes.byref(cnt))
    assert rc == 0 and cnt.value == N
    return sets[:N], ks[:N], N


class Scorer(object):
    """One handle of the C-ABI = one GPU.  Thin: argument marshalling and error mapping only
    (status codes -> ValueError / RuntimeError, SURVEY.md section 8 b)."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        self._h = _vp()
        rc = self._lib.sdpcut_create(int(device_id), ctypes.byref(self._h))
        if rc != 0:
            msg = self._lib.sdpcut_last_error(None).decode()
            self._h = None
            raise SdpCutError("sdpcut_create failed (%d): %s" % (rc, msg))
        self.N = 0
        self.nb_vars = 0
        self.base = 0
        self.round_count = 0     # rounds that wrote the handle's pinned host block (views of it are good until the next one)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc == 0:
            return
        msg = self._lib.sdpcut_last_error(self._h).decode()
        if rc == -1:
            raise ValueError(msg)
        raise SdpCutError("sdpcut error %d: %s" % (rc, msg))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdpcut_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        self._check(self._lib.sdpcut_set_option(self._h, option, int(value)))

    def set_stream(self, stream_ptr):
        """stream_ptr: a hipStream_t as int (0 = the null stream, PyTorch's default stream);
        None restores the handle's own stream."""
        arg = _vp(-1 & 0xFFFFFFFFFFFFFFFF) if stream_ptr is None else (_vp(stream_ptr) if stream_ptr else None)
        self._check(self._lib.sdpcut_set_stream(self._h, arg))

    def get_stat(self, which):
        v = _c.c_int64(0)
        self._check(self._lib.sdpcut_get_stat(self._h, int(which), ctypes.byref(v)))
        return int(v.value)

    def wake(self):
        """an empty kernel on the handle's stream (sdpcut_wake): poke an idle device while the LP solution is still being extracted"""
        self._check(self._lib.sdpcut_wake(self._h))

    def synchronize(self):
        self._check(self._lib.sdpcut_synchronize(self._h))

    # ------------------------------------------------------------------ setup
    def set_network(self, k, widths, params):
        widths = np.ascontiguousarray(widths, dtype=np.int32)
        params = _f64(params)
        self._check(self._lib.sdpcut_set_network(self._h, int(k), widths.shape[0], _ptr(widths, _i32p),
                                                 _ptr(params, _dp), params.shape[0]))

    def set_instance(self, nb_vars, Q_arr):
        Q_arr = _f64(Q_arr)
        if Q_arr.shape != (nb_vars * (nb_vars + 1) // 2,):
            raise ValueError("Q_arr must hold the packed upper triangle, n(n+1)/2 entries")
        self._check(self._lib.sdpcut_set_instance(self._h, int(nb_vars), _ptr(Q_arr, _dp)))
        self.nb_vars = int(nb_vars)

    def set_candidates(self, set_inds, ks, global_base=0):
        set_inds = np.ascontiguousarray(set_inds, dtype=np.int32)
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        if set_inds.ndim != 2 or ks.shape != (set_inds.shape[0],):
            raise ValueError("set_inds must be [N, ld] and ks [N]")
        self._check(self._lib.sdpcut_set_candidates(self._h, set_inds.shape[0], _ptr(set_inds, _i32p),
                                                    set_inds.shape[1], _ptr(ks, _i32p), int(global_base)))
        self.N = int(set_inds.shape[0])
        self.base = int(global_base)
        kmax = int(ks.max()) if ks.size else 2
        self.row_len = kmax * (kmax + 3) // 2

    def set_builtin_networks(self, max_k=5):
        """the reference's four trained MLPs, from the copy compiled into the library"""
        self._check(self._lib.sdpcut_set_builtin_networks(self._h, int(max_k)))

    def set_candidates_philox(self, k, count, seed=7, first_id=0):
        """C4 workload: `count` random k-variable index sets generated on the device, candidate ids
        first_id .. first_id + count - 1 (= the global indices reported)"""
        self._check(self._lib.sdpcut_set_candidates_philox(self._h, int(k), int(count), int(seed), int(first_id)))
        self.N, self.base = int(count), int(first_id)
        self.row_len = int(k) * (int(k) + 3) // 2

    def set_candidates_cover(self, adjacency, dim, max_subs=0):
        """semidefinite vertex cover enumerated on the device into this handle's list
        -> number of candidates (with max_subs > 0 and count >= max_subs the list is NOT replaced)"""
        adj = np.ascontiguousarray(np.asarray(adjacency) != 0, dtype=np.uint8)
        if adj.shape != (self.nb_vars, self.nb_vars):
            raise ValueError("adjacency must be [n, n]")
        cnt = _c.c_int64(0)
        self._check(self._lib.sdpcut_set_candidates_cover(self._h, adj.ctypes.data_as(_c.POINTER(_c.c_uint8)), int(dim),
                                                          int(max_subs or 0), ctypes.byref(cnt)))
        if not (max_subs and cnt.value >= max_subs):
            self.N, self.base = int(cnt.value), 0
            self.row_len = int(dim) * (int(dim) + 3) // 2      # upper bound: the largest size present is <= dim
        return int(cnt.value)

    def set_candidates_cover_split(self, other, adjacency_obj, adjacency_all, dim):
        """The two covers of a QCQP instance (cut_select_qcqp.py:314-334) on the device: this handle gets the sub-problems
        of the cover of `adjacency_all` that also belong to the cover of `adjacency_obj`, `other` the rest
        -> (count of this handle, count of `other`)"""
        ao = np.ascontiguousarray(np.asarray(adjacency_obj) != 0, dtype=np.uint8)
        aa = np.ascontiguousarray(np.asarray(adjacency_all) != 0, dtype=np.uint8)
        if ao.shape != (self.nb_vars, self.nb_vars) or aa.shape != ao.shape:
            raise ValueError("adjacency must be [n, n]")
        n_in, n_out = _c.c_int64(0), _c.c_int64(0)
        u8 = _c.POINTER(_c.c_uint8)
        self._check(self._lib.sdpcut_set_candidates_cover_split(self._h, other._h, ao.ctypes.data_as(u8), aa.ctypes.data_as(u8), int(dim),
                                                                ctypes.byref(n_in), ctypes.byref(n_out)))
        for sc, n in ((self, n_in.value), (other, n_out.value)):
            sc.N, sc.base = int(n), 0
            sc.row_len = int(dim) * (int(dim) + 3) // 2
        return int(n_in.value), int(n_out.value)

    def get_candidates(self, local_idx):
        """-> (set_inds int32 [count, 5] padded with -1, ks int32 [count]) of candidates by local index"""
        self.drop_pending()      # (lists that live on the device are read through their scorer whenever somebody indexes them)
        idx = np.ascontiguousarray(local_idx, dtype=np.int64)
        out = np.empty((max(idx.shape[0], 1), 5), dtype=np.int32)
        ks = np.empty(max(idx.shape[0], 1), dtype=np.int32)
        self._check(self._lib.sdpcut_get_candidates(self._h, idx.shape[0], _ptr(idx, _i64p), _ptr(out, _i32p), _ptr(ks, _i32p)))
        return out[:idx.shape[0]], ks[:idx.shape[0]]

    def set_point(self, vars_values):
        vv = _f64(vars_values)
        n = self.nb_vars
        if vv.shape != (n * (n + 1) // 2 + n,):
            raise ValueError("vars_values must be [X packed | x] of length n(n+1)/2 + n")
        self._check(self._lib.sdpcut_set_point(self._h, _ptr(vv, _dp)))

    def point_buffer(self):
        """The handle's pinned staging block for the LP point as a numpy array of n(n+1)/2 + n doubles: fill it in place
        and pass IT to set_point / select_round / round_csr -- the library then skips its host copy of the point."""
        p = _dp()
        self._check(self._lib.sdpcut_point_buffer(self._h, ctypes.byref(p)))
        n = self.nb_vars
        count = n * (n + 1) // 2 + n
        addr = ctypes.cast(p, _vp).value
        return np.frombuffer((_c.c_double * count).from_address(addr), dtype=np.float64, count=count)

    def set_point_device(self, dev_ptr):
        self._check(self._lib.sdpcut_set_point_device(self._h, _vp(dev_ptr)))

    # ------------------------------------------------------------------ hot path
    def score(self, flags):
        self._check(self._lib.sdpcut_score(self._h, int(flags)))

    def get_scores(self, eig=True, obj=True):
        e = np.empty(self.N) if eig else None
        o = np.empty(self.N) if obj else None
        self._check(self._lib.sdpcut_get_scores(self._h, _ptr(e, _dp), _ptr(o, _dp)))
        return e, o

    def rank(self, strat, sel_size=0, max_out=None):
        """-> (idx int64[w], score float64[w], n_total, new_strat, counters dict)"""
        cap = self.N if max_out is None else max(0, min(int(max_out), self.N))
        idx = np.empty(max(cap, 1), dtype=np.int64)
        sc = np.empty(max(cap, 1), dtype=np.float64)
        n_total = _c.c_int64(0)
        new_strat = _c.c_int32(0)
        cnt = np.zeros(4, dtype=np.int64)
        self._check(self._lib.sdpcut_rank(self._h, int(strat), int(sel_size), cap, _ptr(idx, _i64p), _ptr(sc, _dp),
                                          ctypes.byref(n_total), ctypes.byref(new_strat), _ptr(cnt, _i64p)))
        w = min(cap, n_total.value)
        counters = dict(nb_violated=int(cnt[0]), strong=int(cnt[1]), violated=int(cnt[2]), nb_positive=int(cnt[3]))
        return idx[:w], sc[:w], int(n_total.value), int(new_strat.value), counters

    def rank_fetch(self, offset, count):
        idx = np.empty(max(count, 1), dtype=np.int64)
        sc = np.empty(max(count, 1), dtype=np.float64)
        self._check(self._lib.sdpcut_rank_fetch(self._h, int(offset), int(count), _ptr(idx, _i64p), _ptr(sc, _dp)))
        return idx[:count], sc[:count]

    def rank_device(self, strat, sel_size, max_out, d_idx_ptr, d_score_ptr):
        n_written, n_total, new_strat = _c.c_int64(0), _c.c_int64(0), _c.c_int32(0)
        cnt = np.zeros(4, dtype=np.int64)
        self._check(self._lib.sdpcut_rank_device(self._h, int(strat), int(sel_size), int(max_out), _vp(d_idx_ptr),
                                                 _vp(d_score_ptr), ctypes.byref(n_written), ctypes.byref(n_total),
                                                 ctypes.byref(new_strat), _ptr(cnt, _i64p)))
        counters = dict(nb_violated=int(cnt[0]), strong=int(cnt[1]), violated=int(cnt[2]), nb_positive=int(cnt[3]))
        return int(n_written.value), int(n_total.value), int(new_strat.value), counters

    def merge_topk_device(self, count, d_scores_ptr, d_ids_ptr, max_out, d_score_out_ptr, d_id_out_ptr,
                          d_secondary_ptr=None):
        self._check(self._lib.sdpcut_merge_topk_device(
            self._h, int(count), _vp(d_scores_ptr), _vp(d_secondary_ptr) if d_secondary_ptr else None,
            _vp(d_ids_ptr), int(max_out), _vp(d_score_out_ptr), _vp(d_id_out_ptr)))

    def gather_scores_device(self, count, d_ids_ptr, d_eig_out_ptr=None, d_obj_out_ptr=None):
        self._check(self._lib.sdpcut_gather_scores_device(
            self._h, int(count), _vp(d_ids_ptr), _vp(d_eig_out_ptr) if d_eig_out_ptr else None,
            _vp(d_obj_out_ptr) if d_obj_out_ptr else None))

    def cut_rows(self, local_idx):
        idx = np.ascontiguousarray(local_idx, dtype=np.int64)
        c = idx.shape[0]
        lam = np.empty(c)
        coef = np.empty((c, ROW_LD))
        rhs = np.empty(c)
        cols = np.empty((c, ROW_LD), dtype=np.int64)
        ks = np.empty(c, dtype=np.int32)
        self._check(self._lib.sdpcut_cut_rows(self._h, c, _ptr(idx, _i64p), _ptr(lam, _dp), _ptr(coef, _dp),
                                              _ptr(rhs, _dp), _ptr(cols, _i64p), _ptr(ks, _i32p)))
        return lam, coef, rhs, cols, ks

    def select_round(self, strat, sel_size, copy=True, point=None):
        """score (if needed) + rank + cut rows of the head in one call
        -> dict(idx, score, lam, coef, rhs, ks, n_total, new_strat, counters).

        point: the LP point [X packed | x] of this round -- set_point and the round in ONE library call
        (sdpcut_round_view).

        The device writes the results into a pinned host block owned by the handle
        (sdpcut_select_round_view).  copy=False returns numpy views of that block: no host copy
        at all, valid until the next call on this Scorer; copy=True (default) detaches them."""
        ld = self.row_len
        self.round_count += 1
        st = getattr(self, "_sr_out", None)
        if st is None:      # the out-parameters of the call, made once per Scorer (a round is a few microseconds of host time)
            block, cap, n_out, n_total, new_strat = _c.c_void_p(), _c.c_int64(0), _c.c_int64(0), _c.c_int64(0), _c.c_int32(0)
            cnt = np.zeros(4, dtype=np.int64)
            st = self._sr_out = (block, cap, n_out, n_total, new_strat, cnt,
                                 (ctypes.byref(block), ctypes.byref(cap), ctypes.byref(n_out), ctypes.byref(n_total), ctypes.byref(new_strat),
                                  _ptr(cnt, _i64p)))
        block, cap, n_out, n_total, new_strat, cnt, refs = st
        if point is not None:
            vv = _f64(point)
            n = self.nb_vars
            if vv.shape != (n * (n + 1) // 2 + n,):
                raise ValueError("vars_values must be [X packed | x] of length n(n+1)/2 + n")
            self._check(self._lib.sdpcut_round_view(self._h, _ptr(vv, _dp), int(strat), int(sel_size), ld, *refs))
        else:
            self._check(self._lib.sdpcut_select_round_view(self._h, int(strat), int(sel_size), ld, *refs))
        w, c = int(n_out.value), int(cap.value)
        if block.value and c:
            key = (block.value, c, ld)
            if getattr(self, "_round_view_key", None) != key:       # the block is reused round after round
                nbytes = 64 + c * 8 * (4 + ld) + c * 4
                buf = (_c.c_char * nbytes).from_address(block.value)
                o = 64
                v_idx = np.frombuffer(buf, dtype=np.int64, count=c, offset=o); o += 8 * c
                v_sc = np.frombuffer(buf, dtype=np.float64, count=c, offset=o); o += 8 * c
                v_lam = np.frombuffer(buf, dtype=np.float64, count=c, offset=o); o += 8 * c
                v_rhs = np.frombuffer(buf, dtype=np.float64, count=c, offset=o); o += 8 * c
                v_coef = np.frombuffer(buf, dtype=np.float64, count=c * ld, offset=o).reshape(c, ld); o += 8 * c * ld
                v_ks = np.frombuffer(buf, dtype=np.int32, count=c, offset=o)
                self._round_views, self._round_view_key = (v_idx, v_sc, v_lam, v_rhs, v_coef, v_ks), key
            if getattr(self, "_round_slices_w", None) != (key, w):      # (the head usually has the same length round after round)
                self._round_slices, self._round_slices_w = tuple(a[:w] for a in self._round_views), (key, w)
            idx, sc, lam, rhs, coef, ks = self._round_slices
            if copy:
                idx, sc, lam, rhs, coef, ks = (a.copy() for a in (idx, sc, lam, rhs, coef, ks))
        else:
            idx, sc, lam, rhs = np.empty(0, np.int64), np.empty(0), np.empty(0), np.empty(0)
            coef, ks = np.empty((0, ld)), np.empty(0, np.int32)
        return dict(idx=idx, score=sc, lam=lam, coef=coef, rhs=rhs, ks=ks,
                    n_total=int(n_total.value), new_strat=int(new_strat.value),
                    counters=dict(nb_violated=int(cnt[0]), strong=int(cnt[1]), violated=int(cnt[2]),
                                  nb_positive=int(cnt[3])))

    def round_csr(self, strat, sel_size, point=None, copy=False):
        """One round with the cuts assembled on the device (sdpcut_round_csr): LP point -> score -> rank -> eigen-cuts
        of the head as ONE CSR block -> dict(idx, score, lam, ks, set_inds [., 5], n_total, new_strat, counters,
        row_entry, indptr, indices, values, rhs).  The arrays are numpy views of the handle's pinned host block (written
        by the device, valid until the next call on this Scorer); copy=True detaches them.  point=None keeps the
        current LP point."""
        vv = self._csr_point(point)
        self.round_count += 1
        out = self._csr_out()
        self._check(self._lib.sdpcut_round_csr(self._h, _ptr(vv, _dp), int(strat), int(sel_size), ctypes.byref(out)))
        return self._csr_unpack(out, copy)

    def round_csr_begin(self, strat, sel_size, point=None):
        """First half of round_csr (sdpcut_round_csr_begin): enqueue the whole round, do not wait.  Several Scorers may begin
        before any ends; their device work overlaps."""
        vv = self._csr_point(point)
        self.round_count += 1
        self._check(self._lib.sdpcut_round_csr_begin(self._h, _ptr(vv, _dp), int(strat), int(sel_size)))
        self.pending = object()          # identifies THIS begun round until it is ended (or dropped)
        return self.pending

    def round_csr_end(self, copy=False):
        """Second half of round_csr (sdpcut_round_csr_end): wait for the round begun on this Scorer; the same dict."""
        out = self._csr_out()
        self.pending = None
        self._check(self._lib.sdpcut_round_csr_end(self._h, ctypes.byref(out)))
        return self._csr_unpack(out, copy)

    pending = None

    def drop_pending(self):
        """end a begun round nobody will collect (its results are discarded); True if there was one"""
        if self.pending is None:
            return False
        self.round_csr_end()
        return True

    def _csr_point(self, point):
        if point is None:
            return None
        vv = _f64(point)
        n = self.nb_vars
        if vv.shape != (n * (n + 1) // 2 + n,):
            raise ValueError("vars_values must be [X packed | x] of length n(n+1)/2 + n")
        return vv

    def _csr_out(self):
        out = getattr(self, "_csr_struct", None)
        if out is None:
            out = self._csr_struct = RoundCsr()
        return out

    def _csr_unpack(self, out, copy):
        c, w, r = int(out.cap), int(out.n_out), int(out.n_rows)
        if c and out.idx:
            key = (out.idx, c, int(out.row_ld))
            if getattr(self, "_csr_view_key", None) != key:       # the block is reused round after round
                ld = int(out.row_ld)

                def view(ptr, dtype, count, shape=None):
                    a = np.frombuffer((_c.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype, count=count)
                    return a.reshape(shape) if shape else a
                self._csr_views = dict(
                    idx=view(out.idx, np.int64, c), score=view(out.score, np.float64, c), lam=view(out.lam_min, np.float64, c),
                    ks=view(out.ks, np.int32, c), set_inds=view(out.set_inds, np.int32, 5 * c, (c, 5)),
                    row_entry=view(out.row_entry, np.int32, c), indptr=view(out.indptr, np.int32, c + 1),
                    indices=view(out.indices, np.int32, c * ld), values=view(out.values, np.float64, c * ld),
                    rhs=view(out.rhs, np.float64, c))
                self._csr_view_key = key
            v = self._csr_views
            nnz = int(out.nnz)
            res = dict(idx=v["idx"][:w], score=v["score"][:w], lam=v["lam"][:w], ks=v["ks"][:w], set_inds=v["set_inds"][:w],
                       row_entry=v["row_entry"][:r], indptr=v["indptr"][:r + 1], indices=v["indices"][:nnz],
                       values=v["values"][:nnz], rhs=v["rhs"][:r])
            if copy:
                res = {k: a.copy() for k, a in res.items()}
        else:
            z = np.zeros
            res = dict(idx=z(0, np.int64), score=z(0), lam=z(0), ks=z(0, np.int32), set_inds=z((0, 5), np.int32),
                       row_entry=z(0, np.int32), indptr=z(1, np.int32), indices=z(0, np.int32), values=z(0), rhs=z(0))
        cnt = out.counters
        res.update(n_total=int(out.n_total), new_strat=int(out.new_strat),
                   counters=dict(nb_violated=int(cnt[0]), strong=int(cnt[1]), violated=int(cnt[2]), nb_positive=int(cnt[3])))
        return res

    # ------------------------------------------------------------------ sharded round (multi-GPU)
    def shard_head_device(self, strat, count, d_record_ptr):
        """enqueue this shard's packed head record (8 + 2*count int64 words); no host sync"""
        self._check(self._lib.sdpcut_shard_head_device(self._h, int(strat), int(count), _vp(d_record_ptr)))

    def _shard_views(self, block, w, m, ld):
        key = (block, w, m, ld)
        if getattr(self, "_shard_view_key", None) != key:           # the block is reused round after round
            nbytes = w * 64 + m * 8 * (4 + ld) + m * 8
            buf = (_c.c_char * nbytes).from_address(block)
            o = 0
            hdr = np.frombuffer(buf, dtype=np.int64, count=w * 8, offset=o).reshape(w, 8); o += w * 64
            idx = np.frombuffer(buf, dtype=np.int64, count=m, offset=o); o += 8 * m
            sc = np.frombuffer(buf, dtype=np.float64, count=m, offset=o); o += 8 * m
            lam = np.frombuffer(buf, dtype=np.float64, count=m, offset=o); o += 8 * m
            rhs = np.frombuffer(buf, dtype=np.float64, count=m, offset=o); o += 8 * m
            coef = np.frombuffer(buf, dtype=np.float64, count=m * ld, offset=o).reshape(m, ld); o += 8 * m * ld
            ks = np.frombuffer(buf, dtype=np.int32, count=m, offset=o); o += 4 * m
            pos = np.frombuffer(buf, dtype=np.int32, count=m, offset=o)
            self._shard_views_cache = dict(headers=hdr, idx=idx, score=sc, lam=lam, coef=coef, rhs=rhs, ks=ks, pos=pos)
            self._shard_view_key = key
        return self._shard_views_cache

    def shard_finish_enqueue(self, world, count, d_allrec_ptr, sel_size, fields=2, pitch_words=0):
        """second half of a sharded round, enqueued without host synchronisation (sdpcut_shard_finish_enqueue);
        fields = 3: the records carry obj_improve as secondary key (SDPCUT_PART_COMBALL); pitch_words: distance between
        consecutive ranks' records when several lists share the gathered buffer (0 = one list per buffer)"""
        # (the library refuses a second enqueue -- or any other stateful call -- before the wait: state changes only on success)
        self._check(self._lib.sdpcut_shard_finish_enqueue(self._h, int(world), int(count), int(fields), _vp(d_allrec_ptr),
                                                          int(pitch_words), int(sel_size), self.row_len))
        self.round_count += 1
        self._shard_pending = (int(world), int(sel_size), self.row_len)

    def shard_finish_wait(self, own=True):
        """-> dict(headers, idx, score, lam, coef, rhs, ks[, pos, n_own]): views of the handle's pinned block, which the
        device wrote (valid until the next round on this Scorer); own=True: lam / coef / rhs / ks hold only this shard's
        n_own rows, compacted in head order, pos[:n_own] their positions in the head"""
        w, m, ld = self._shard_pending
        block, n_own = _c.c_void_p(), _c.c_int64(0)
        self._check(self._lib.sdpcut_shard_finish_wait(self._h, 1 if own else 0, ctypes.byref(block), ctypes.byref(n_own)))
        out = dict(self._shard_views(block.value, w, m, ld))
        if own:
            out["n_own"] = int(n_own.value)
        else:
            out.pop("pos")
        return out

    def shard_finish_round(self, world, count, d_allrec_ptr, sel_size, copy=False):
        """merge the gathered records, cut rows of this shard's entries
        -> dict(headers [world, 8], idx, score, lam, coef, rhs, ks), each of sel_size entries:
        numpy views of the handle's pinned host block, which the device wrote directly
        (valid until the next call on this Scorer; copy=True detaches them)"""
        m, ld, w = int(sel_size), self.row_len, int(world)
        self.round_count += 1
        block = _c.c_void_p()
        self._check(self._lib.sdpcut_shard_finish_round_view(
            self._h, w, int(count), _vp(d_allrec_ptr), m, ld, ctypes.byref(block)))
        out = {k: v for k, v in self._shard_views(block.value, w, m, ld).items() if k != "pos"}
        return {k: v.copy() for k, v in out.items()} if copy else out

    def shard_finish_round_own(self, world, count, d_allrec_ptr, sel_size):
        """like shard_finish_round, but lam / coef / rhs / ks hold only the n_own rows of this shard
        (compacted in head order by the library) and pos[:n_own] their positions in the head
        -> dict(headers, idx, score, lam, coef, rhs, ks, pos, n_own); views, see above"""
        m, ld, w = int(sel_size), self.row_len, int(world)
        self.round_count += 1
        block, n_own = _c.c_void_p(), _c.c_int64(0)
        self._check(self._lib.sdpcut_shard_finish_round_own(
            self._h, w, int(count), _vp(d_allrec_ptr), m, ld, ctypes.byref(block), ctypes.byref(n_own)))
        out = dict(self._shard_views(block.value, w, m, ld))
        out["n_own"] = int(n_own.value)
        return out

    # ------------------------------------------------------------------ triangle inequalities
    def tri_preprocess(self, adjacency):
        """-> (triples int32 [T, 3], density uint8 [T]) kept on the device as well."""
        adj = np.ascontiguousarray(np.asarray(adjacency) != 0, dtype=np.uint8)
        if adj.shape != (self.nb_vars, self.nb_vars):
            raise ValueError("adjacency must be [n, n]")
        T = _c.c_int64(0)
        self._check(self._lib.sdpcut_tri_preprocess(self._h, adj.ctypes.data_as(_c.POINTER(_c.c_uint8)),
                                                    ctypes.byref(T)))
        tri = np.empty((max(T.value, 1), 3), dtype=np.int32)
        dens = np.empty(max(T.value, 1), dtype=np.uint8)
        self._check(self._lib.sdpcut_tri_get_triples(self._h, _ptr(tri, _i32p),
                                                     dens.ctypes.data_as(_c.POINTER(_c.c_uint8))))
        self.n_tri = int(T.value)
        return tri[:T.value], dens[:T.value]

    def tri_separate(self, max_out):
        """-> (entry ids 4*triple+type int64[w], violations float64[w], n_violated)"""
        cap = max(0, min(int(max_out), 4 * getattr(self, "n_tri", 0)))
        ent = np.empty(max(cap, 1), dtype=np.int64)
        vio = np.empty(max(cap, 1))
        nv, nw = _c.c_int64(0), _c.c_int64(0)
        self._check(self._lib.sdpcut_tri_separate(self._h, cap, _ptr(ent, _i64p), _ptr(vio, _dp), ctypes.byref(nv),
                                                  ctypes.byref(nw)))
        return ent[:nw.value], vio[:nw.value], int(nv.value)

    def eig_batch(self, k, x_rho, X_rho, want_vectors=False):
        x_rho, X_rho = _f64(x_rho), _f64(X_rho)
        c = x_rho.shape[0]
        if x_rho.shape != (c, k) or X_rho.shape != (c, k * (k + 1) // 2):
            raise ValueError("x_rho must be [count, k] and X_rho [count, k(k+1)/2]")
        w = np.empty((c, k + 1))
        v = np.empty((c, k + 1, k + 1)) if want_vectors else None
        self._check(self._lib.sdpcut_eig_batch(self._h, int(k), c, _ptr(x_rho, _dp), _ptr(X_rho, _dp), _ptr(w, _dp),
                                               _ptr(v, _dp)))
        return (w, v) if want_vectors else w

    def nn_batch(self, k, inputs):
        inputs = _f64(inputs)
        c = inputs.shape[0]
        if inputs.shape != (c, k * (k + 3) // 2):
            raise ValueError("inputs must be [count, k(k+3)/2]")
        out = np.empty(c)
        self._check(self._lib.sdpcut_nn_batch(self._h, int(k), c, _ptr(inputs, _dp), _ptr(out, _dp)))
        return out

    def last_timing(self):
        ms = np.zeros(2)
        self._check(self._lib.sdpcut_last_timing(self._h, _ptr(ms, _dp), 2))
        return float(ms[0]), float(ms[1])

    def mfma_probe(self, A, B):
        A, B = _f64(A), _f64(B)
        assert A.shape == (16, 4) and B.shape == (4, 16)
        C = np.empty((16, 16))
        self._check(self._lib.sdpcut_mfma_probe(self._h, _ptr(A, _dp), _ptr(B, _dp), _ptr(C, _dp)))
        return C

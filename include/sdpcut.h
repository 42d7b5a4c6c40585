/*
 * sdpcut.h -- C-ABI of the MI355X (gfx950) cut-scoring library  libsdpcut_hip.so
 *
 * Drop-in boundary for ONE hot path of rb2309/SDPCutSel-via-NN: the per-round scoring,
 * ranking and generation of low-dimensional PSD cuts.  The reference reaches native code
 * for this path through exactly one FFI:
 *
 *     ctypes.cdll.LoadLibrary('neural_nets/NNs.so')            cut_select_qp.py:297
 *     double neural_net_{2,3,4,5}D(const double X[d(d+3)/2])   cut_select_qp.py:299-303, 579-582
 *
 * i.e. one scalar MLP evaluation per candidate per call, with the gather, the eigen-
 * decomposition (numpy/LAPACK, cut_select_qp.py:788-797), the scoring arithmetic
 * (:573-582), the ranking (:601-654) and the cut rows (:737-750) done in Python around it.
 * This library replaces that per-candidate FFI by a batched, handle-based one that runs
 * the whole per-round computation on the GPU.  Each entry point below names the reference
 * lines it replaces.  Plain C types only: caller-owned contiguous host buffers unless a
 * parameter is named d_* (device pointer).  Every function returns 0 on success or a
 * negative SDPCUT_E* code; sdpcut_last_error() returns the message.  A handle is not
 * re-entrant; different handles are independent (one per GPU / per thread).
 *
 * There is NO CPU fallback: every entry point needs a gfx950 device and fails with
 * SDPCUT_ENODEVICE otherwise.
 */
#ifndef SDPCUT_H
#define SDPCUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is ALL it exports (tests/test_host_cpu.py) */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct sdpcut_ctx *sdpcut_handle;

enum {
    SDPCUT_OK = 0,
    SDPCUT_EINVAL = -1,    /* bad argument (mirrors the reference's asserts, cut_select_qp.py:91-94) */
    SDPCUT_ENODEVICE = -2, /* no usable gfx950 device */
    SDPCUT_EHIP = -3,      /* HIP runtime error */
    SDPCUT_ESTATE = -4,    /* call order violated (e.g. score before set_point) */
    SDPCUT_ENOMEM = -5
};

/* score flags */
enum { SDPCUT_EIG = 1, SDPCUT_NN = 2 };

/* selection strategies, numbering of cut_select_algo (cut_select_qp.py:79-80) */
enum { SDPCUT_STRAT_FEAS = 1, SDPCUT_STRAT_OPT = 2, SDPCUT_STRAT_COMB = 4 };
/* Partial ranking for candidate sets sharded over several GPUs (SURVEY.md section 8 e): only
 * the "strong" class of the combined scan (cut_select_qp.py:607-613: obj_improve > 0 and
 * violated), by obj_improve.  Its merged per-shard heads give the global position at which
 * the reference's scan stops.  n_total / counters[0] = size of the class. */
enum { SDPCUT_PART_STRONG = 104 };
/* The other regime of the combined scan over shards: fewer than sel_size strong candidates exist OVERALL, so the scan
 * visits every entry (cut_select_qp.py:606-623) and each shard's own combined ranking is a sub-list of the global one.
 * sdpcut_shard_head_device with this value packs the shard's head by (new score, obj_improve, id) with obj_improve as a
 * third field of the record (the merge's secondary key: the second stable sort of :625 keeps the order of the first, :601). */
enum { SDPCUT_PART_COMBALL = 105 };

/* kernel variants for sdpcut_set_option(SDPCUT_OPT_KERNEL, ...) */
enum { SDPCUT_KERNEL_MFMA = 0, SDPCUT_KERNEL_SIMPLE = 1, SDPCUT_KERNEL_VALU = 2 };
/* SDPCUT_OPT_FUSE_KEYS (default 1): sdpcut_select_round lets the score kernels count their scores by
 * the leading radix digit of the selection keys (per workgroup in LDS, flushed with no-return atomics)
 * together with the violated / positive counters; the top-k selection then starts at its second digit
 * and builds the keys from the scores while it reads them, so the separate key pass over the scores
 * (17.5 us per round at 10^6 candidates) disappears.  0: the selection runs its own first pass. */
/* SDPCUT_OPT_AUTO_REGIME (default 1): sdpcut_select_round with the combined strategy lets the score
 * kernels count the strong candidates and the selection pick its regime on the device (one selection,
 * no host round trip whether or not sel_size strong candidates exist).  0: the selection assumes the
 * common regime and the host repeats it when the count says otherwise (the round-1 behaviour). */
/* SDPCUT_OPT_FUSED_TAIL (default 1): the top-k selection runs its later passes, the count and the
 * compaction in one launch behind bounded grid barriers instead of four launches; 0 = one launch per
 * pass (A/B and fallback). */
/* SDPCUT_OPT_COOP_LAUNCH (default 0): make that launch a cooperative one (hipLaunchCooperativeKernel), so
 * that the runtime guarantees the co-residency its grid barriers rely on.  Measured on MI355X: +20 us per
 * round (0.455 instead of 0.435 ms).  Without it the grid is capped at one workgroup per CU (256 x 256
 * threads, 38 KB of LDS each), which an otherwise idle device always holds, and every wait is bounded: a
 * barrier that does not complete within a few milliseconds voids the selection and the full-sort path
 * answers (counted by SDPCUT_STAT_SELECT_FALLBACKS). */
/* SDPCUT_OPT_EIG_KERNEL (default 1): launches that compute only lambda_min (feasibility rounds,
 * cut_select_qp.py:639-654) run the dedicated eigenvalue kernel -- one launch over all size classes, compiled
 * without the MLP's register state -- instead of the scoring kernels' eigenvalue branch (0: A/B). */
/* SDPCUT_OPT_ONE_LAUNCH (default 1): a list with several size classes is scored by ONE launch in which every class has its own
 * range of workgroups (the shipped networks; same scores bit for bit as a launch per class).  0, or a list the launch does not
 * cover (user networks that need the clamped path, the VALU / simple kernel variants): a launch per class, see the next option. */
/* SDPCUT_OPT_SIDE_STREAMS (default 0 since r4 -- the default configuration picks no code path from a timing; only lists that
 * SDPCUT_OPT_ONE_LAUNCH does not cover get here at all): a list with several size classes may score its smaller classes on side streams of the handle,
 * between a fork and a join event, next to the largest class on the handle's stream.  0: one launch after the other; 1: side
 * streams; 2: the first multi-class scoring of a candidate list measures both forms (~1 ms, once) and keeps the faster --
 * whether the streams run side by side depends on which hardware queues the process's streams were given. */
/* SDPCUT_OPT_STREAM_PRIORITY (default 0): 1 re-creates the handle's own stream with the device's highest priority.  For a handle
 * whose list is SHORT and whose rounds run next to another handle's (the QCQP round's objective cover beside its constraints
 * cover, sdpcut_round_csr_begin): its few small kernels are then dispatched ahead of the other list's waiting workgroups instead of
 * behind them.  Not allowed while a round is pending; ignored by a handle that runs on a caller's stream (sdpcut_set_stream). */
/* SDPCUT_OPT_PREFILTER (default 1; needs SDPCUT_OPT_FUSE_KEYS): the eigenvalue kernel (every class) and the score kernel (3-variable
 * classes) of a fused round also count the class members in a 2048-bin window over the top seventeen bits of their selection keys
 * (csrc/topk_dev.h: one LDS atomic per candidate; a workgroup reports, when it retires, only the bins down to the one that holds
 * its own m-th largest member, and publishes that floor).  Where the bin of the k-th largest key lies above every floor and its
 * members fit the sort buffers, the selection is resolved from the table: one pass over the scores, no digit pass, no grid barrier.
 * 0: the radix passes of rounds 2-4 (A/B; identical results). */
enum { SDPCUT_OPT_KERNEL = 1, SDPCUT_OPT_TIMING = 2, SDPCUT_OPT_FUSE_KEYS = 3, SDPCUT_OPT_AUTO_REGIME = 4,
       SDPCUT_OPT_FUSED_TAIL = 5, SDPCUT_OPT_COOP_LAUNCH = 6, SDPCUT_OPT_EIG_KERNEL = 7, SDPCUT_OPT_STREAM_PRIORITY = 8,
       SDPCUT_OPT_SIDE_STREAMS = 9, SDPCUT_OPT_ONE_LAUNCH = 10, SDPCUT_OPT_PREFILTER = 11 };

/* Counters of a handle: SDPCUT_STAT_ROUNDS = fused rounds served (sdpcut_select_round*),
 * SDPCUT_STAT_SELECT_FALLBACKS = rounds whose radix selection declared itself void (a grid barrier
 * timed out because other work kept its workgroups from starting) and were answered by the full-sort
 * path instead -- same result, ~1 ms instead of ~0.1 ms.
 * SDPCUT_STAT_SCORED = the measures (SDPCUT_EIG | SDPCUT_NN) scored at the current point.
 * SDPCUT_STAT_TIE_SPLITS (r4) = every-entry-visited combined rankings whose threshold group of EQUAL new scores
 * did not fit the sort buffers (structured LP vertices) and was cut by its secondary key -- obj_improve, then index,
 * cut_select_qp.py:601 under :625 -- with two more radix selections; until round 3 these rounds were fallbacks.
 * SDPCUT_STAT_DIRECT_SELECTIONS (r5) = selections resolved from the fine histogram the score kernels leave (no digit pass, no grid
 *   barrier: SDPCUT_OPT_PREFILTER); the others ran the radix passes (short lists, masses of equal keys at the threshold, the
 *   every-entry-visited regime).  Read from the device: the call waits for the handle's stream.  Same results either way.
 * SDPCUT_STAT_PF_BIN / _PF_FLOOR / _PF_COUNT: what the last selection that looked at the fine histogram found there -- the fine bin of
 *   the k-th largest key (-1: none), the floor the producers published (fine bins), the members at or above that bin.  Diagnostics. */
enum { SDPCUT_STAT_ROUNDS = 1, SDPCUT_STAT_SELECT_FALLBACKS = 2, SDPCUT_STAT_SCORED = 3, SDPCUT_STAT_TIE_SPLITS = 4,
       SDPCUT_STAT_DIRECT_SELECTIONS = 5, SDPCUT_STAT_PF_BIN = 6, SDPCUT_STAT_PF_FLOOR = 7, SDPCUT_STAT_PF_COUNT = 8 };
int sdpcut_get_stat(sdpcut_handle h, int which, int64_t *value);

/* Maximum sub-problem size (assert dim <= 5, cut_select_qp.py:93) */
#define SDPCUT_MAX_K 5
/* row stride of the padded coefficient output of sdpcut_cut_rows: k + k(k+1)/2 <= 20 */
#define SDPCUT_ROW_LD 20

int sdpcut_version(void);
const char *sdpcut_last_error(sdpcut_handle h); /* h may be NULL: error of the last failed create */

/* Lifetime.  Replaces _load_neural_nets' LoadLibrary (cut_select_qp.py:284-303). */
int sdpcut_create(int device_id, sdpcut_handle *out);
int sdpcut_destroy(sdpcut_handle h);
int sdpcut_set_option(sdpcut_handle h, int option, int64_t value);
/* Run all work of this handle on an existing HIP stream (hipStream_t passed as void*).
 * NULL is the HIP null stream (what PyTorch calls its default stream);
 * SDPCUT_OWN_STREAM restores the handle's own non-blocking stream. */
#define SDPCUT_OWN_STREAM ((void *)(intptr_t)-1)
int sdpcut_set_stream(sdpcut_handle h, void *hip_stream);
int sdpcut_synchronize(sdpcut_handle h);
/* (r5) Enqueue an empty kernel on the handle's stream and return at once.  In the reference's loop a separation round follows an LP
 * solve of 0.1-10 s (cut_select_qp.py:149-200, :193-200) during which the device falls idle; a round issued to an idle device costs
 * 0.1-0.2 ms more than one issued back to back (bench.py: secondary.cold_round).  A caller that pokes the device as soon as its
 * solver returns -- before it extracts the solution vector -- gets 40-65 us of that back.  Optional; changes no result. */
int sdpcut_wake(sdpcut_handle h);

/*
 * Trained MLP for k-variable candidates (replaces the constants baked into NNs.so;
 * neural_nets/neural_net_kD.m constants section).  n_layers counts the linear output
 * layer; widths[l] = outputs of layer l (last = 1).  params is packed as
 *   xoffset[d_in], gain[d_in], ymin,
 *   for each layer: W[width][fan_in] row-major, b[width],
 *   y_ymin, y_gain, y_xoffset                      with d_in = k(k+3)/2.
 */
int sdpcut_set_network(sdpcut_handle h, int k, int n_layers, const int32_t *widths,
                       const double *params, int64_t n_params);

/* The four trained MLPs of the reference (k = 2 .. max_k), compiled into the library from
 * data/nn_weights.npz: sdpcut_set_network for each of them without the caller holding weights. */
int sdpcut_set_builtin_networks(sdpcut_handle h, int max_k);

/* Instance table: packed row-major upper triangle of the objective, length n(n+1)/2
 * (self._Q_arr, cut_select_qp.py:318-321 / cut_select_qcqp.py:247-256). */
int sdpcut_set_instance(sdpcut_handle h, int32_t nb_vars, const double *Q_arr);

/*
 * Candidate index sets (self._agg_list[i][0], cut_select_qp.py:529-540).  set_inds is
 * [N][ld] int32 with the first ks[i] entries of row i valid (2 <= ks[i] <= 5, ld >= max k);
 * Xarr_inds, Q_slice and max_elem are re-derived on the device from Q_arr.
 * global_base is added to every candidate index this handle reports (multi-GPU shards).
 */
int sdpcut_set_candidates(sdpcut_handle h, int64_t N, const int32_t *set_inds, int32_t ld,
                          const int32_t *ks, int64_t global_base);

/*
 * The C4 workload of SURVEY.md section 8 d, generated in device memory: N random k-variable index
 * sets, candidate id -> set through a counter-based generator (Philox4x32-10 keyed by `seed`,
 * counter = (id, attempt, block); k draws floor(u32 * nb_vars / 2^32), sorted, redrawn until
 * distinct: every k-subset equally likely, ids independent of each other and of N).  The handle's
 * list becomes the ids first_id .. first_id + N - 1, which are also the GLOBAL indices it reports
 * (shard r of a multi-GPU run passes first_id = r * N).  Replaces, for synthetic batches, the
 * host-built list of sdpcut_set_candidates; csrc/philox.h is the arithmetic, synthetic.py its
 * numpy twin.  Needs nb_vars >= 2 k.
 */
int sdpcut_set_candidates_philox(sdpcut_handle h, int32_t k, int64_t N, uint64_t seed, int64_t first_id);

/*
 * Semidefinite vertex cover P^E_dim enumerated on the device, straight into the handle's candidate
 * list (replaces _get_sdp_vertex_cover's index-set loops, cut_select_qp.py:399-524, AND the upload
 * of their result): same sets, same order as sdpcut_enumerate_cover.  adjacency as there
 * ([nb_vars][nb_vars] bytes, nb_vars of sdpcut_set_instance, <= 1024).  *count_out = number of
 * candidates.  max_subs > 0 mirrors the reference's RAM guard (_THRES_MAX_SUBS, :117-120): with
 * count >= max_subs only the count is returned and the handle's list is left alone; 0 = no guard.
 */
int sdpcut_set_candidates_cover(sdpcut_handle h, const uint8_t *adjacency, int32_t dim, int64_t max_subs,
                                int64_t *count_out);

/*
 * The two candidate lists of a QCQP instance (replaces __get_vertex_cover, cut_select_qcqp.py:314-334, including its
 * list-membership intersection): h_in receives the sub-problems of the cover of `adjacency_all` (objective + all
 * constraints) that also belong to the cover of `adjacency_obj` (self._agg_list, :331), h_out the others
 * (agg_list_cons, :332-333), both in the order of the `adjacency_all` enumeration.  Everything happens on the device
 * (two enumerations, one binary search per set, prefix sums); the handles must sit on the same device and hold the
 * same instance (sdpcut_set_instance).  Adjacencies as in sdpcut_set_candidates_cover.
 */
int sdpcut_set_candidates_cover_split(sdpcut_handle h_in, sdpcut_handle h_out, const uint8_t *adjacency_obj,
                                      const uint8_t *adjacency_all, int32_t dim, int64_t *n_in, int64_t *n_out);

/* Index sets of `count` candidates given by LOCAL index, device -> host: set_inds_out [count][5]
 * padded with -1, ks_out [count] (0 for an index outside the list).  For lists that were generated
 * or enumerated on the device: the host names only the few thousand selected candidates. */
int sdpcut_get_candidates(sdpcut_handle h, int64_t count, const int64_t *idx, int32_t *set_inds_out,
                          int32_t *ks_out);

/* LP point vars_values = [X packed (L) | x (n)]  (cut_select_qp.py:137, 200, 547). */
int sdpcut_set_point(sdpcut_handle h, const double *vars_values);
/* The handle's pinned, device-mapped staging block for the LP point: *buf = (L + n) doubles the caller may fill IN PLACE
 * (let the LP solver write its solution there) and then pass to sdpcut_set_point / sdpcut_round_view / sdpcut_round_csr,
 * which recognise the pointer and skip their host copy (4 MB at n = 1000: ~110 us per round).  Valid until the next
 * sdpcut_set_instance; write to it only between rounds (a round's completion means its point has left the host). */
int sdpcut_point_buffer(sdpcut_handle h, double **buf);
/* same, from a device buffer (async copy on the handle's stream) */
int sdpcut_set_point_device(sdpcut_handle h, const void *d_vars_values);

/*
 * Score every candidate at the current point (replaces the per-candidate loop bodies of
 * _sel_eigcut_by_ordering_on_measure, cut_select_qp.py:570-582 and :642-648):
 *   SDPCUT_EIG: eigmin[i]      = lambda_min([[1, x^T],[x, X]])          (a6; numpy.linalg.eigvalsh(...)[0] at cut_select_qp.py:796.
 *               (r4) Householder tridiagonalisation + Laguerre's iteration, Jacobi for nearly multiple lambda_min (csrc/lmin.h):
 *               as far from the exact eigenvalue as LAPACK is, ~1e-16 on average, <= 2e-15 against LAPACK on matrices of norm 2-4)
 *   SDPCUT_NN : obj_improve[i] = (-S) * max_elem + nn([x | Q_slice]) * max_elem  (a4, a5)
 * Results stay on the device; fetch with sdpcut_get_scores.
 */
int sdpcut_score(sdpcut_handle h, uint32_t flags);
int sdpcut_get_scores(sdpcut_handle h, double *eigmin, double *obj_improve); /* either may be NULL */

/*
 * Rank (replaces the sorts / combined scan, cut_select_qp.py:601-632 and :649-654).
 *   strat 1: violated candidates (lambda_min < -1e-15) by -lambda_min descending
 *   strat 2: all candidates by obj_improve descending
 *   strat 4: combined scan with BIG_M, using sel_size; *new_strat = 1 or 4 (:630-631)
 * Ties keep ascending candidate index (Python's stable sort).  Writes the first
 * min(max_out, length) entries: idx_out = global candidate index, score_out = ranking score.
 * *n_total = full length of the reference's list (N, or nb_violated for strat 1).
 * counters (may be NULL) = {nb_violated, strong_violated, violated_in_scan, nb_positive}.
 * Requires a preceding sdpcut_score with the flags the strategy needs.
 */
int sdpcut_rank(sdpcut_handle h, int strat, int64_t sel_size, int64_t max_out,
                int64_t *idx_out, double *score_out, int64_t *n_total,
                int32_t *new_strat, int64_t *counters);
/* same, writing to device buffers (for the multi-GPU all-gather); returns counts on host */
int sdpcut_rank_device(sdpcut_handle h, int strat, int64_t sel_size, int64_t max_out,
                       void *d_idx_out, void *d_score_out, int64_t *n_written,
                       int64_t *n_total, int32_t *new_strat, int64_t *counters);

/* Read entries [offset, offset+count) of the ranking produced by the last sdpcut_rank /
 * sdpcut_rank_device call (the reference hands back the whole sorted list; callers normally
 * consume only its head, so the tail stays on the device until asked for). */
int sdpcut_rank_fetch(sdpcut_handle h, int64_t offset, int64_t count, int64_t *idx_out,
                      double *score_out);

/*
 * Order `count` entries by (score descending, secondary descending, id ascending) and write
 * the first max_out: the replicated merge after the all-gather of per-shard top-k (SURVEY
 * 8 e).  d_secondary may be NULL (two-level key).  The secondary key carries obj_improve for
 * the combined strategy, whose second stable sort keeps first-sort order among equal new
 * scores (cut_select_qp.py:601, :625).  All pointers are device pointers; ids are int64.
 */
int sdpcut_merge_topk_device(sdpcut_handle h, int64_t count, const void *d_scores,
                             const void *d_secondary, const void *d_ids, int64_t max_out,
                             void *d_score_out, void *d_id_out);

/* eigmin / obj_improve of `count` candidates given by GLOBAL index, device to device
 * (either output may be NULL). */
int sdpcut_gather_scores_device(sdpcut_handle h, int64_t count, const void *d_ids,
                                void *d_eig_out, void *d_obj_out);

/*
 * Eigen-cut rows of selected candidates (replaces the loop body of _gen_eigcuts_selected,
 * cut_select_qp.py:737-750).  idx are LOCAL candidate indices (global - global_base).
 *   lam_min[c]                 smallest eigenvalue
 *   coef[c*SDPCUT_ROW_LD + .]  [2v0v1..2v0vk | v1^2, 2v1v2, .., vk^2] with |v_i|<=1e-15 zeroed
 *   rhs[c]                     -v0^2
 *   cols[c*SDPCUT_ROW_LD + .]  [L + i for i in set_inds] + Xarr_inds   (int64)
 *   ks[c]                      candidate size k (row length = k + k(k+1)/2)
 * A row is only meaningful when lam_min < -1e-15 (the reference skips the others).
 * v = unit eigenvector of lam_min (numpy.linalg.eigh at cut_select_qp.py:796-797): inverse iteration with lam_min (LU of
 * A - lam I, residual a few ulp of ||A||) -- the value the handle holds for the candidate at the current point (SDPCUT_EIG scored,
 * e.g. by a feasibility / combined round) or, (r4) when it holds none, the value the same solver computes on the spot
 * (Householder + Laguerre, csrc/lmin.h).  Whenever lam_min is multiple or within 1e-10 ||A|| of the next eigenvalue, where only an
 * eigenSPACE is defined, v comes from a Jacobi iteration with vectors.  Either way the row is a unit
 * eigenvector's cut; two solvers agree on it to eps / gap, as the reference's LAPACK and this library always did.
 */
int sdpcut_cut_rows(sdpcut_handle h, int64_t count, const int64_t *idx, double *lam_min,
                    double *coef, double *rhs, int64_t *cols, int32_t *ks);

/*
 * One selection round in one call: what the loop body does between two LP solves
 * (cut_select_qp.py:165-182: _sel_eigcut_by_ordering_on_measure followed by
 * _gen_eigcuts_selected).  Scores with the flags the strategy needs (unless already scored
 * at this point), ranks, and produces the eigen-cut rows of the first min(sel_size, length)
 * entries; a single device-to-host transfer returns everything.  Outputs are sized for
 * sel_size entries; *n_out entries are written.  idx_out are GLOBAL candidate indices;
 * the column indices of row c follow from its index set (see sdpcut_cut_rows) and are not
 * transferred.  lam_min / coef / rhs / ks as in sdpcut_cut_rows, except that coef rows have
 * the caller's stride coef_ld (>= k + k(k+1)/2 of the largest candidate, <= SDPCUT_ROW_LD):
 * a 3-variable-only list moves 9 instead of 20 doubles per row over PCIe.
 */
int sdpcut_select_round(sdpcut_handle h, int strat, int64_t sel_size, int32_t coef_ld,
                        int64_t *idx_out, double *score_out, double *lam_min, double *coef,
                        double *rhs, int32_t *ks, int64_t *n_out, int64_t *n_total,
                        int32_t *new_strat, int64_t *counters);
/*
 * Zero-copy form: the device writes the round's results straight into a pinned host block owned
 * by the handle; *block points at it and stays valid until the next call on the handle.  With
 * cap = *cap_out = min(sel_size, N) the block is
 *     64 bytes reserved | int64 idx[cap] | double score[cap] | double lam_min[cap] |
 *     double rhs[cap] | double coef[cap][coef_ld] | int32 ks[cap]
 * of which the first *n_out entries of every array are meaningful (*block is NULL if cap = 0).
 */
int sdpcut_select_round_view(sdpcut_handle h, int strat, int64_t sel_size, int32_t coef_ld,
                             const void **block, int64_t *cap_out, int64_t *n_out,
                             int64_t *n_total, int32_t *new_strat, int64_t *counters);

/*
 * One cutting-plane round from the LP point to the cut rows in ONE call: sdpcut_set_point(vars_values)
 * followed by sdpcut_select_round_view(...) -- what the separation step of cut_select_qp.py:165-182
 * does between two LP solves.  Same results as the two calls; the point transfer and the score kernel are
 * enqueued back to back and the caller crosses the FFI once per round.
 */
int sdpcut_round_view(sdpcut_handle h, const double *vars_values, int strat, int64_t sel_size, int32_t coef_ld,
                      const void **block, int64_t *cap_out, int64_t *n_out,
                      int64_t *n_total, int32_t *new_strat, int64_t *counters);

/*
 * One cutting-plane round with the cuts ASSEMBLED: everything the separation step of cut_select_qp.py:165-182 hands
 * to the LP -- _sel_eigcut_by_ordering_on_measure (:543-703), _gen_eigcuts_selected (:705-755) and the row objects
 * the latter builds one cut at a time (cplex.SparsePair(ind=[L + i for i in set_inds] + Xarr_inds, val=coeffs),
 * rhs -v0^2, sense "G"; :744-750) -- as one block of rows in compressed sparse row form, assembled on the device
 * and stored straight into a pinned host block owned by the handle (SURVEY.md section 8 f row 4).
 *
 * vars_values: the LP point [X packed | x] (sdpcut_set_point is part of the call), or NULL to keep the current one.
 * sel_size: the strategy's quota AND the number of head entries returned (cap = min(sel_size, N)).
 * All pointers of *out point INTO the handle's block: valid until the next call on the handle, never freed by the caller.
 *   head (n_out entries, rank order):  idx (global candidate ids), score, lam_min (NaN for a candidate of another
 *       shard), ks (candidate size), set_inds [.][5] (index sets padded with -1)
 *   cuts (n_rows <= n_out; an entry yields one iff its lam_min < -1e-15, :739), in head order:
 *       row_entry[r] = head position of cut r;  rhs[r];  indptr[r] .. indptr[r + 1] = its span in indices / values
 *       (n_rows + 1 entries, indptr[n_rows] = nnz);  indices = LP columns (:747),  values = coefficients (:745-746).
 * The cuts of the first m head entries are rows 0 .. r-1 with r = #{row_entry < m} (row_entry ascends), i.e. a
 * prefix of the block: a caller that consumes fewer entries (strong_only, :725-726) slices, nothing is recomputed.
 */
typedef struct sdpcut_round_csr {
    int64_t cap, n_out, n_total;
    int32_t new_strat;
    int32_t row_ld;                 /* longest possible row (k + k(k+1)/2 of the largest candidate size): indices / values hold cap * row_ld */
    int64_t counters[4];            /* as sdpcut_rank */
    const int64_t *idx;
    const double *score;
    const double *lam_min;
    const int32_t *ks;
    const int32_t *set_inds;
    int64_t n_rows, nnz;
    const int32_t *row_entry;
    const int32_t *indptr;
    const int32_t *indices;
    const double *values;
    const double *rhs;
} sdpcut_round_csr_t;
int sdpcut_round_csr(sdpcut_handle h, const double *vars_values, int strat, int64_t sel_size, sdpcut_round_csr_t *out);
/* The same in two halves: _begin enqueues the whole round and returns without waiting, _end waits and fills *out (and runs the
 * general ranking path itself in the rare cases the enqueued selection is not the answer).  A handle holds one pending round;
 * DIFFERENT handles may all begin before any ends -- their device work overlaps.  The QCQP round ranks two lists per LP point
 * (cut_select_qcqp.py:64-78): begin(objective cover), begin(constraints cover), end, end. */
int sdpcut_round_csr_begin(sdpcut_handle h, const double *vars_values, int strat, int64_t sel_size);
int sdpcut_round_csr_end(sdpcut_handle h, sdpcut_round_csr_t *out);

/*
 * The same round over candidate shards (one handle per GPU, SURVEY 8 e): the two device-side
 * halves around the single all-gather the caller performs (torch.distributed / RCCL).
 *
 * sdpcut_shard_head_device: enqueue, WITHOUT host synchronisation, this shard's head of the
 * ranking (strat 1, 2, SDPCUT_PART_STRONG or SDPCUT_PART_COMBALL) into one packed device record of
 * 8 + fields * count int64 words (fields = 3 for SDPCUT_PART_COMBALL, else 2)
 *     [list length, nb_violated, nb_positive, entries written, void flag, 0, 0, 0 |
 *      count scores (fp64 bits) | count GLOBAL ids | (fields = 3) count secondary keys = obj_improve (fp64 bits)]
 * padded with (-inf, INT64_MAX, -inf); 1 <= count <= 16384.  Measures the strategy needs and sdpcut_score has
 * not computed since the last sdpcut_set_point are scored by this call (a sharded round is set_point,
 * shard_head, all-gather, shard_finish); when none of them has been, the score kernels also prepare the
 * selection's first radix digit (SDPCUT_OPT_FUSE_KEYS).
 *
 * sdpcut_shard_finish_enqueue / sdpcut_shard_finish_wait: d_allrec holds the `world` records in rank order
 * (`fields` as above).  Enqueue merges them by (score descending, [secondary descending,] id ascending) -- the order
 * of the reference's stable sorts on one list (cut_select_qp.py:601, :625, :653) --, keeps the first sel_size entries
 * and produces the eigen-cut rows of those that belong to THIS shard (the others: ks = 0, lam_min = NaN), all stored
 * by the device into the handle's pinned block; wait returns the block (layout below) once it is complete -- the
 * round's only host synchronisation.  Several handles may have their halves enqueued before the first wait.
 * sdpcut_shard_finish_round*: the two calls in one, fields = 2.  headers [world][8] are the record headers (the
 * caller sums them); entries beyond the summed list length are pads.
 */
int sdpcut_shard_head_device(sdpcut_handle h, int strat, int64_t count, void *d_record);
/* pitch_words: distance between the records of consecutive ranks in d_allrec, in int64 words (0 = the record's own length:
 * one list per gathered buffer); larger when the records of several lists travel in one all-gather (the QCQP round's two
 * covers: d_allrec then points at this list's record of rank 0 inside the buffer). */
int sdpcut_shard_finish_enqueue(sdpcut_handle h, int32_t world, int64_t count, int32_t fields, const void *d_allrec,
                                int64_t pitch_words, int64_t sel_size, int32_t coef_ld);
/* compact_own != 0: block as sdpcut_shard_finish_round_own, *n_own = rows of this shard; 0: as sdpcut_shard_finish_round_view */
int sdpcut_shard_finish_wait(sdpcut_handle h, int32_t compact_own, const void **block, int64_t *n_own);
int sdpcut_shard_finish_round(sdpcut_handle h, int32_t world, int64_t count,
                              const void *d_allrec, int64_t sel_size, int32_t coef_ld,
                              int64_t *headers_out, int64_t *idx_out, double *score_out,
                              double *lam_min, double *coef, double *rhs, int32_t *ks);
/* Zero-copy form (see sdpcut_select_round_view): *block = the handle's pinned host block
 *     int64 headers[world][8] | int64 idx[sel_size] | double score[sel_size] |
 *     double lam_min[sel_size] | double rhs[sel_size] | double coef[sel_size][coef_ld] |
 *     int32 ks[sel_size]
 * written by the device, valid until the next call on the handle. */
int sdpcut_shard_finish_round_view(sdpcut_handle h, int32_t world, int64_t count,
                                   const void *d_allrec, int64_t sel_size, int32_t coef_ld,
                                   const void **block);
/* Same block, but the *n_own rows of THIS shard are moved to the front of lam_min / rhs / coef /
 * ks (keeping head order) and an array  int32 pos[sel_size]  is appended behind ks: pos[j] = the
 * position in the merged head of own row j.  idx / score stay the full replicated head. */
int sdpcut_shard_finish_round_own(sdpcut_handle h, int32_t world, int64_t count,
                                  const void *d_allrec, int64_t sel_size, int32_t coef_ld,
                                  const void **block, int64_t *n_own);

/*
 * Batched twin of _get_eigendecomp (cut_select_qp.py:788-797) for explicit sub-matrices:
 * x_rho [count][k], X_rho [count][k(k+1)/2] (upper triangle, row-major).  Writes ascending
 * eigenvalues [count][k+1] and, if evecs != NULL, eigenvectors [count][k+1][k+1] with
 * evecs[c][i][j] = component i of eigenvector j (numpy's column convention).
 */
int sdpcut_eig_batch(sdpcut_handle h, int k, int64_t count, const double *x_rho,
                     const double *X_rho, double *eigvals, double *evecs);

/* Batched raw MLP forward: inputs [count][d_in] -> out [count] (the NNs.so call, batched). */
int sdpcut_nn_batch(sdpcut_handle h, int k, int64_t count, const double *inputs, double *out);

/* Timing of the last sdpcut_score / sdpcut_rank (HIP events on the handle's stream).
 * SDPCUT_OPT_TIMING = 1: events around the score kernels only; = 2: also around the ranking.
 * ms[0] = score kernels, ms[1] = rank (-1 when not recorded). */
int sdpcut_last_timing(sdpcut_handle h, double *ms, int n);

/*
 * Semidefinite vertex cover P^E_dim (replaces the index-set enumeration of
 * _get_sdp_vertex_cover, cut_select_qp.py:399-524, ch_ext = 0): every clique of size `dim`
 * of the sparsity graph plus every maximal clique of size 2..dim-1, in the reference's order.
 * adjacency is [nb_vars][nb_vars] bytes (non-zero = edge; symmetrised, diagonal ignored).
 * Writes at most max_out rows of set_inds_out [.][5] (padded with -1) / ks_out and always the
 * full count (call with max_out = 0 to size the arrays; the reference bails out at 4e6,
 * _THRES_MAX_SUBS).  Host function, no handle needed.
 */
int sdpcut_enumerate_cover(int32_t nb_vars, const uint8_t *adjacency, int32_t dim, int64_t max_out,
                           int32_t *set_inds_out, int32_t *ks_out, int64_t *count_out);

/*
 * Triangle inequalities (SURVEY.md section 8 f row 3; cut_select_qp.py:799-863).
 * sdpcut_tri_preprocess (replaces __preprocess_triangle_ineq, :799-822): keeps the triples
 *   i1<i2<i3 with at least two of their three edges in the sparsity graph (adjacency as in
 *   sdpcut_enumerate_cover), lexicographic; needs sdpcut_set_instance.  sdpcut_tri_get_triples
 *   copies them out ([T][3], density 2 or 3 per triple).
 * sdpcut_tri_separate (replaces the scan and sort of __separate_and_add_triangle, :829-842):
 *   at the current LP point computes the four violations of every triple, keeps those
 *   >= 1e-7, orders them by (density desc, violation desc), ties in entry order, and returns
 *   the first max_out as entry ids 4*triple + type with their violations.  *n_violated is the
 *   length of the full list (the caller derives the number of cuts from it, :844-845).
 */
int sdpcut_tri_preprocess(sdpcut_handle h, const uint8_t *adjacency, int64_t *n_triples);
int sdpcut_tri_get_triples(sdpcut_handle h, int32_t *triples_out, uint8_t *density_out);
int sdpcut_tri_separate(sdpcut_handle h, int64_t max_out, int64_t *entry_out, double *viol_out,
                        int64_t *n_violated, int64_t *n_written);

/* Self-test hook: multiplies A[16x4] * B[4x16] with v_mfma_f64_16x16x4_f64 using the
 * fragment maps the MLP kernel assumes; C row-major [16][16]. */
int sdpcut_mfma_probe(sdpcut_handle h, const double *A, const double *B, double *C);

/*
 * The reference's own FFI (neural_net_{2,3,4,5}D, NNs_initialize, NNs_terminate; cut_select_qp.py:297-303) is exported by the
 * sibling library libsdpcut_nns.so -- include/sdpcut_nns.h -- which binds to this one privately (sdpcut_create,
 * sdpcut_set_builtin_networks, sdpcut_nn_batch).  Until round 4 this library exported the six names itself.
 */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SDPCUT_H */

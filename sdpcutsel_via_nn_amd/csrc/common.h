// Shared host-side declarations of libsdpcut_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/sdpcut.h"

#define SDPCUT_NEG_EIGVAL (-1e-15) /* _THRES_NEG_EIGVAL, cut_select_qp.py:24 */
#define SDPCUT_BIG_M 1000.0        /* _BIG_M, cut_select_qp.py:26 */

// lists shorter than this do not count the fine histogram of their selection (topk_dev.h): their heads come from the one-workgroup
// selections or from passes over a few thousand keys, and their epilogues are too small to zero 34 KB more
#define SDPCUT_PF_MIN_N 32768

#define MAX_HIDDEN 64
#define MAX_LAYERS 5

// Device-side description of one trained MLP (k = 2..5); passed to kernels by value.
struct NetDev {
    int d_in;
    int n_hidden;          // hidden (tansig) layers
    int width;             // hidden width (all hidden layers equal: 64 or 50)
    int s0, sh;            // k-steps (of 4) of the first / the other hidden layers
    // The MFMA kernel's copies of the hidden-layer weights and biases are pre-scaled by -1/4: the
    // accumulators then hold y/8 = -n/4 of tansig's exp(-2n) = exp(y) directly (exact: scaling by
    // a power of two commutes with every rounding of the dot product), one multiply less per
    // activation.
    const double *wfrag;   // MFMA A-fragments (x -1/4): layer-major, then [t][s][lane]
    const double *wtail;   // (x -1/4) [layer][4][64]: rows 48..51 of each hidden layer (VALU tail of the MFMA kernel)
    const double *bias_q;  // (x -1/4) [n_hidden][64] zero padded
    const double *wvalu;   // scalar-operand packing: layer-major, then [j/8][i][j%8], zero padded
    const double *bias;    // [n_hidden][64] zero padded
    const double *wout;    // [64] zero padded output weights
    const double *inmap;   // xoffset[d_in] | gain[d_in]
    const double *raw_w[MAX_LAYERS]; // row-major [out][in] (simple kernel)
    const double *raw_b[MAX_LAYERS];
    double ymin, b_out, y_ymin, y_gain, y_xoffset;
    // every pre-activation is provably below the tansig's overflow clamp when the mapped inputs lie in
    // [-SDPCUT_INPUT_CLAMP, SDPCUT_INPUT_CLAMP] (sdpcut_set_network): the MFMA kernel then clamps the 9..20
    // inputs of a candidate once instead of its 150..256 activations
    int unclamped_ok;
};
#define SDPCUT_INPUT_CLAMP 3.0

struct Bucket {
    int64_t n = 0;
    int32_t *d_set = nullptr;   // SoA [k][n]
    int32_t *d_orig = nullptr;  // [n] local candidate index (position in caller's list)
};

struct NetHost {
    bool set = false;
    NetDev dev{};
    double *d_blob = nullptr;   // one allocation behind all device pointers of dev
};

// a fused round between its two halves (capi.hip: round_begin / round_end)
struct PendingRound {
    bool active = false, csr = false, fast_tried = false;
    int strat = 0;
    int32_t ld = 0;
    int64_t sel_size = 0, cap = 0, serial = 0;
};

// every entry point that touches the handle's scores, staging or pinned block refuses to run between the two halves of a round
// (fused: round_csr_begin / _end; sharded: shard_finish_enqueue / _wait -- their kernels are still writing d_stage and the pinned block)
#define SDPCUT_NO_PENDING(h)                                                                                              \
    do {                                                                                                                  \
        if ((h)->pend.active)                                                                                             \
            return sdpcut_fail((h), SDPCUT_ESTATE, "a round begun with sdpcut_round_csr_begin is pending on this handle: end it first"); \
        if ((h)->shard_pending_serial)                                                                                    \
            return sdpcut_fail((h), SDPCUT_ESTATE, "a sharded round enqueued with sdpcut_shard_finish_enqueue is pending on this handle: " \
                                                   "sdpcut_shard_finish_wait first");                                    \
    } while (0)

struct sdpcut_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // the small size classes of a mixed cover are scored on side streams next to the largest one (launch_score)
    bool one_launch = true;      // SDPCUT_OPT_ONE_LAUNCH: the size classes of a mixed list in one launch (score_mfma_all_kernel)
    int side_streams = 0;        // SDPCUT_OPT_SIDE_STREAMS: 0 off (default since r4: no code path picked by a timing), 1 on, 2 measured once per candidate list (side_choice)
    int side_choice = -1;        // -1 not measured yet, 0 one after the other, 1 side streams
    float side_ms[2] = {0.f, 0.f};
    hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    std::string err;
    int kernel_variant = SDPCUT_KERNEL_MFMA;
    bool fuse_keys = true;         // SDPCUT_OPT_FUSE_KEYS (see include/sdpcut.h)
    bool auto_regime = true;       // SDPCUT_OPT_AUTO_REGIME
    bool fused_tail = true;        // SDPCUT_OPT_FUSED_TAIL
    bool eig_kernel = true;        // SDPCUT_OPT_EIG_KERNEL: eigenvalue-only launches run eig_only_kernel (eig.hip)
    bool pf_counted = false;       // the scoring launches of the current round all counted the fine histogram (launch_score)
    bool prefilter = true;         // SDPCUT_OPT_PREFILTER: fine histogram in the score kernels, direct selection (topk_dev.h)
    unsigned long long *d_stats = nullptr;   // device counters that outlive a round: [0] direct selections
    bool coop_launch = false;      // SDPCUT_OPT_COOP_LAUNCH: cooperative launch of the kernels with grid barriers (+20 us per round)
    int64_t stat_rounds = 0, stat_fallbacks = 0, stat_tie_splits = 0;   // sdpcut_get_stat
    int timing = 0;                // 0 off, 1 events around the score kernel, 2 also around the ranking
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timed_score = false;      // ev[0] / ev[1] were attached to the last score launch
    float ms_score = 0.f, ms_rank = 0.f;
    int n_cu = 256;

    int32_t nb_vars = 0;
    int64_t L = 0;
    double *d_Q = nullptr;      // [L]
    double *d_vars = nullptr;   // [L + n]
    // set by sdpcut_shard_head_device around its selection: the sort's last kernel also writes the
    // record header and the padding (topk.hip, tk_mergerank_kernel)
    int64_t *shard_rec = nullptr;
    int64_t shard_rec_count = 0, shard_rec_len = -1;
    bool have_point = false;

    int64_t N = 0, base = 0;
    int32_t row_len_max = 5;       // k + k(k+1)/2 of the largest candidate size present
    Bucket bucket[SDPCUT_MAX_K + 1];
    int32_t *d_set_orig = nullptr; // [N][5] padded, caller order
    int32_t *d_k = nullptr;        // [N]
    double *d_eig = nullptr, *d_obj = nullptr; // [N] caller order
    uint32_t scored = 0;
    int64_t last_total = -1;       // length of the last ranking (-1: none), see sdpcut_rank_fetch

    NetHost net[SDPCUT_MAX_K + 1];

    // ranking workspace (sized to N by ensure_rank_ws)
    int64_t ws_n = 0;              // entries the full-list arrays hold (ensure_rank_ws)
    int64_t key_n = 0;             // entries d_key_a holds (ensure_key_ws)
    uint64_t *d_key_a = nullptr, *d_key_b = nullptr;
    uint32_t *d_val_a = nullptr, *d_val_b = nullptr;
    int32_t *d_flag = nullptr, *d_scan = nullptr;
    int64_t *d_counters = nullptr; // [8]
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    // top-k select workspace (topk.hip)
    void *d_topk_ws = nullptr;
    int64_t tk_coresident = 256;   // workgroups of tk_refine_kernel resident at once (occupancy x CUs), set with the workspace
    void *d_topk_ws_alt = nullptr; // second workspace: zeroed by the epilogue of a round for the next one
    bool topk_alt_clean = false;   // d_topk_ws_alt has been (stream-ordered) zeroed and may be swapped in
    uint64_t *d_sel_key = nullptr;
    uint32_t *d_sel_idx = nullptr;
    // triangle inequalities (tri.hip)
    int32_t *d_tri = nullptr;          // [T][3]
    uint8_t *d_tri_dense3 = nullptr;   // [T] density == 3
    int64_t n_tri = 0;
    std::vector<int32_t> tri_host;
    std::vector<uint8_t> tri_dense_host;
    // small staging
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    void *pinned = nullptr;        // host block of sdpcut_select_round (written by the device, one sync per round)
    void *pinned_dev = nullptr;    // the same memory as the device sees it
    size_t pinned_bytes = 0;
    // sdpcut_set_point: pinned staging copy of the caller's LP point and the event of its transfer
    void *point_stage = nullptr, *point_stage_dev = nullptr;
    size_t point_stage_bytes = 0;
    bool point_inflight = false;   // a transfer out of point_stage may still be running
    int64_t round_serial = 0;      // completion word of the fused round (round_rows_kernel -> pinned header)
    uint32_t *d_done_ticket = nullptr;
    PendingRound pend;
    // sdpcut_shard_finish_enqueue -> sdpcut_shard_finish_wait
    int64_t shard_pending_serial = 0, shard_pending_sel = 0;
    int32_t shard_pending_world = 0, shard_pending_ld = 0;
};

// every host wait on the handle's stream goes through here: it also tells sdpcut_set_point that the
// staging copy of the previous LP point has left the host
static inline hipError_t sdpcut_sync(sdpcut_ctx *h)
{
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) h->point_inflight = false;
    return e;
}

int sdpcut_fail(sdpcut_ctx *h, int code, const std::string &msg);
int ensure_stage(sdpcut_ctx *h, size_t bytes);   // capi.hip: grow h->d_stage
// capi.hip: (re)allocate the device arrays of a list of N candidates, cnt[k] of them of size k
int alloc_candidates(sdpcut_ctx *h, int64_t N, const int64_t cnt[SDPCUT_MAX_K + 1], int64_t global_base);
int ensure_pinned(sdpcut_ctx *h, size_t bytes);  // capi.hip: grow h->pinned / h->pinned_dev
void free_candidates(sdpcut_ctx *h);             // capi.hip: drop the handle's list (N = 0, nothing scored)

#define HIP_TRY(h, expr)                                                                   \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return sdpcut_fail((h), SDPCUT_EHIP,                                           \
                               std::string(#expr) + ": " + hipGetErrorString(e__));        \
    } while (0)

// score.hip
// Optional: the score kernels count their scores by the leading radix digit of the selection keys
// (ScoreArgs::tk): ws = zeroed TopkWs of the selection that follows (topk_begin).
struct ScoreFuse {
    void *ws;
    int mode;      // TK_MODE_FEAS / OPT / STRONG of topk_dev.h: whose keys to count
    int64_t k;     // head size of the selection (the streaming prefilter's bound rises to the k-th largest key); 0: no fine histogram
};
// strong_out (optional, device, 8 int64 replicas): the launches add the number of candidates with
// obj_improve > 0 and lambda_min < -1e-15 to strong_out[workgroup % 8] (needs both flags; used by the device-resolved combined selection)
int launch_score(sdpcut_ctx *h, uint32_t flags, const ScoreFuse *fuse = nullptr, bool *fused = nullptr,
                 int64_t *strong_out = nullptr);
int launch_cut_rows(sdpcut_ctx *h, int64_t count, const int64_t *d_limit, const int64_t *d_idx, int64_t idx_base,
                    double *d_lam, double *d_coef, int coef_ld, double *d_rhs, int64_t *d_cols, int32_t *d_ks);
// Epilogue of a fused round: rows of the ranking head + its ids, scores and the four counters,
// written to `block` (device view of the pinned host block; layout of sdpcut_select_round_view).
int launch_round_rows(sdpcut_ctx *h, int64_t cap, const int64_t *d_c4, const int64_t *d_idx, const double *d_score,
                      int coef_ld, void *block, int64_t hdr_bytes = 64, int64_t done_serial = 0);
int launch_point_copy(sdpcut_ctx *h, const double *src_mapped, int64_t n);
// rows.hip: the round's epilogue in CSR form (sdpcut_round_csr); byte offsets of the block's arrays
struct CsrLayout { size_t idx, score, lam, rhs, values, ks, sets, row_entry, indptr, indices, bytes; };
CsrLayout csr_layout(int64_t cap, int ld);
int launch_round_csr(sdpcut_ctx *h, int64_t cap, const int64_t *d_c4, int64_t limit, const int64_t *d_idx, const double *d_score,
                     int ld, void *block, int64_t serial);
// eig.hip: lambda_min of every candidate, one launch over all size classes; tk = TopkWs of a feasibility selection or NULL
int launch_eig_only(sdpcut_ctx *h, void *tk, hipEvent_t ev_start, hipEvent_t ev_stop, int64_t pf_k = 0);
int wait_round_done(sdpcut_ctx *h, const int64_t *word, int64_t serial);   // capi.hip
int score_for_selection(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t cap, uint32_t need, bool allow_auto, int *stage,
                        bool *auto_out);                                   // capi.hip
int launch_eig_batch(sdpcut_ctx *h, int k, int64_t count, const double *d_x, const double *d_X,
                     double *d_vals, double *d_vecs);
int launch_nn_batch(sdpcut_ctx *h, int k, int64_t count, const double *d_in, double *d_out);
int launch_mfma_probe(sdpcut_ctx *h, const double *d_A, const double *d_B, double *d_C);

// rank.hip
int ensure_rank_ws(sdpcut_ctx *h, int64_t n);
int ensure_key_ws(sdpcut_ctx *h, int64_t n);
int rank_on_device(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, int64_t *d_idx_out,
                   double *d_score_out, int64_t *n_written, int64_t *n_total, int32_t *new_strat,
                   int64_t *counters, int64_t strong_hint = -1);
int merge_topk_on_device(sdpcut_ctx *h, int64_t count, const double *d_scores, const double *d_secondary,
                         const int64_t *d_ids, int64_t max_out, double *d_score_out, int64_t *d_id_out);
int gather_scores_on_device(sdpcut_ctx *h, int64_t count, const int64_t *d_ids, double *d_eig_out, double *d_obj_out);
int rank_fetch_on_device(sdpcut_ctx *h, int64_t offset, int64_t count, int64_t *d_idx_out, double *d_score_out);
void free_rank_ws(sdpcut_ctx *h);

// stage: see topk_select_enqueue; auto_regime: the combined strategy's regime is resolved on the device
// from the strong count the score kernels of this round left in the selection workspace
int rank_fast_enqueue(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, int64_t *d_idx_out,
                      double *d_score_out, const int64_t **d_c4, int stage = 0, bool auto_regime = false);
// TK_MODE_* the fast path would use for this request, 0 if it is not eligible
int rank_fast_mode(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, double *score_add);
int rank_fast_finish(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, const int64_t c4[7],
                     int64_t *n_written, int64_t *n_total, int32_t *new_strat, int64_t *counters_out);

// tri.hip
int tri_preprocess(sdpcut_ctx *h, const uint8_t *adjacency, int64_t *n_triples);
int tri_separate(sdpcut_ctx *h, int64_t max_out, int64_t *d_entry_out, double *d_viol_out, int64_t *n_violated,
                 int64_t *n_written);

// topk.hip
int topk_select_enqueue(sdpcut_ctx *h, int mode, int64_t k, double score_add, int64_t *d_idx_out,
                        double *d_score_out, const int64_t **d_counters_out, int stage = 0, int64_t sel = 0);
// allocate / zero (or swap in the pre-zeroed) workspace of the next selection and return it together
// with the key array: what a score launch needs to run the selection's first pass itself
// (ScoreFuse); follow with topk_select_enqueue(..., stage = 3)
int topk_begin(sdpcut_ctx *h, void **ws, uint64_t **keys);
bool topk_fuse_ok(const sdpcut_ctx *h, int64_t k, bool comb);     // may a score launch take a ScoreFuse for a head of k entries?
int64_t *topk_strong_counter(void *ws);      // TopkWs::strong_rep (TK_SREP = 8 replicas) of a workspace handed out by topk_begin
int topk_select_on_device(sdpcut_ctx *h, int mode, int64_t k, double score_add, int64_t *d_idx_out,
                          double *d_score_out, int64_t cnt[5]);
int topk_select_keys_on_device(sdpcut_ctx *h, int64_t n, int64_t k, int64_t *d_idx_out, double *d_val_out, int64_t cnt[5]);
void free_topk_ws(sdpcut_ctx *h);
// the void COMBALL selection in the handle's workspace, answered by two selections over precomputed keys (see topk.hip)
int topk_tie_split(sdpcut_ctx *h, int64_t k, int64_t *d_idx_out, double *d_score_out, int64_t *k_eff_out);
// the workspace NOT used by the selection enqueued last, and its size in 8-byte words: a later
// kernel of the same stream may zero it and then set h->topk_alt_clean (saves the next memset)
int topk_alt_ws(sdpcut_ctx *h, uint64_t **ptr, int *words);

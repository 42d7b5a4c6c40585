// extern "C" surface of libsdpcut_hip.so (declared in include/sdpcut.h).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

#include "common.h"

static std::string g_create_err;
static std::mutex g_mu;

int sdpcut_fail(sdpcut_ctx *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    else {
        std::lock_guard<std::mutex> lk(g_mu);
        g_create_err = msg;
    }
    return code;
}

// Host wait for the serial number the last workgroup of a round's epilogue stores (system scope, after
// its results) into pinned memory.  Bounded: after ~2 s without the word the stream is synchronised
// the ordinary way, which also surfaces a faulted kernel as an error.
int wait_round_done(sdpcut_ctx *h, const int64_t *word, int64_t serial)
{
    const volatile int64_t *w = (const volatile int64_t *)word;
    for (long spin = 0; spin < 400000000L; ++spin) {
        if (*w == serial) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            h->point_inflight = false;      // the round ran behind the point's transfer
            return 0;
        }
        __builtin_ia32_pause();
        if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(h->stream) != hipErrorNotReady) break;
    }
    HIP_TRY(h, sdpcut_sync(h));
    if (*w != serial) return sdpcut_fail(h, SDPCUT_EHIP, "round epilogue did not report completion");
    return 0;
}

int ensure_stage(sdpcut_ctx *h, size_t bytes)
{
    if (bytes <= h->stage_bytes) return 0;
    hipFree(h->d_stage);
    h->d_stage = nullptr;
    h->stage_bytes = 0;
    HIP_TRY(h, hipMalloc(&h->d_stage, bytes));
    h->stage_bytes = bytes;
    return 0;
}

// Pinned host block the device can also write (kernel stores = the device-to-host transfer).
int ensure_pinned(sdpcut_ctx *h, size_t bytes)
{
    if (h->pinned_bytes >= bytes) return 0;
    HIP_TRY(h, sdpcut_sync(h));
    if (h->pinned) (void)hipHostFree(h->pinned);
    h->pinned = nullptr;
    h->pinned_dev = nullptr;
    h->pinned_bytes = 0;
    HIP_TRY(h, hipHostMalloc(&h->pinned, bytes, hipHostMallocMapped));
    HIP_TRY(h, hipHostGetDevicePointer(&h->pinned_dev, h->pinned, 0));
    std::memset(h->pinned, 0, bytes < 64 ? bytes : 64);     // header incl. the completion word of wait_round_done
    h->pinned_bytes = bytes;
    return 0;
}


// Device arrays of a candidate list of N entries, cnt[k] of them with k variables (the callers
// fill them: sdpcut_set_candidates from host arrays, the Philox generator and the cover
// enumeration on the device).  Frees the previous list; sizes the ranking workspace.
int alloc_candidates(sdpcut_ctx *h, int64_t N, const int64_t cnt[SDPCUT_MAX_K + 1], int64_t global_base)
{
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, sdpcut_sync(h));
    free_candidates(h);
    h->base = global_base;
    const size_t nn = (size_t)(N < 1 ? 1 : N);
    HIP_TRY(h, hipMalloc((void **)&h->d_set_orig, nn * 5 * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc((void **)&h->d_k, nn * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc((void **)&h->d_eig, nn * sizeof(double)));
    HIP_TRY(h, hipMalloc((void **)&h->d_obj, nn * sizeof(double)));
    h->row_len_max = 5;
    for (int k = 2; k <= SDPCUT_MAX_K; ++k) {
        Bucket &b = h->bucket[k];
        b.n = cnt[k];
        if (!cnt[k]) continue;
        h->row_len_max = k * (k + 3) / 2;
        HIP_TRY(h, hipMalloc((void **)&b.d_set, (size_t)cnt[k] * k * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc((void **)&b.d_orig, (size_t)cnt[k] * sizeof(int32_t)));
    }
    h->N = N;
    return 0;      // (the ranking workspaces are allocated by whoever first needs them: ensure_key_ws / ensure_rank_ws)
}

extern "C" {

int sdpcut_version(void) { return 200; }

const char *sdpcut_last_error(sdpcut_handle h)
{
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> lk(g_mu);
    return g_create_err.c_str();
}

int sdpcut_create(int device_id, sdpcut_handle *out)
{
    if (!out) return sdpcut_fail(nullptr, SDPCUT_EINVAL, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return sdpcut_fail(nullptr, SDPCUT_ENODEVICE,
                           "no HIP device visible: this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return sdpcut_fail(nullptr, SDPCUT_EINVAL, "device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
        return sdpcut_fail(nullptr, SDPCUT_EHIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return sdpcut_fail(nullptr, SDPCUT_ENODEVICE,
                           std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    sdpcut_ctx *h = new (std::nothrow) sdpcut_ctx();
    if (!h) return sdpcut_fail(nullptr, SDPCUT_ENOMEM, "out of host memory");
    h->device = device_id;
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&h->d_counters, 8 * sizeof(int64_t)) != hipSuccess ||
        hipMalloc((void **)&h->d_stats, 4 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(h->d_stats, 0, 4 * sizeof(unsigned long long)) != hipSuccess) {
        delete h;
        return sdpcut_fail(nullptr, SDPCUT_EHIP, "stream / counter allocation failed");
    }
    h->stream = h->own_stream;
    for (int i = 0; i < 4; ++i) hipEventCreate(&h->ev[i]);
    *out = h;
    return SDPCUT_OK;
}

} // extern "C"

void free_candidates(sdpcut_ctx *h)
{
    for (int k = 0; k <= SDPCUT_MAX_K; ++k) {
        hipFree(h->bucket[k].d_set);
        hipFree(h->bucket[k].d_orig);
        h->bucket[k] = Bucket();
    }
    hipFree(h->d_set_orig); hipFree(h->d_k); hipFree(h->d_eig); hipFree(h->d_obj);
    h->d_set_orig = nullptr; h->d_k = nullptr; h->d_eig = nullptr; h->d_obj = nullptr;
    h->N = 0; h->scored = 0; h->last_total = -1;
    h->topk_alt_clean = false;      // (how much of the selection workspace a round's epilogue zeroes depends on the list's length)
    h->side_choice = -1;      // (a new list measures for itself whether its small size classes go to side streams)
}

// Scores of a round's selection (strategy strat, head of `cap` entries), not computed yet at this point:
// *stage / *auto_out are what topk_select_enqueue / rank_fast_enqueue take.
// If nothing has been scored at the point and the head comes from the radix select, the score kernels work
// for the selection that follows.  Combined strategy (allow_auto): they count the strong candidates into
// its workspace and the selection resolves its regime on the device (at least sel_size strong ones:
// those, + BIG_M; fewer: every entry visited) -- one selection and no host round trip in either regime.
// SDPCUT_OPT_FUSE_KEYS: they also count the class members by the leading digit of the selection keys, so
// that the selection starts at its second digit and needs no key pass (the every-entry-visited regime of
// the combined strategy runs its own first digit inside the selection).
int score_for_selection(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t cap, uint32_t need, bool allow_auto, int *stage,
                        bool *auto_out)
{
    *stage = 0;
    *auto_out = false;
    int rc;
    const int fast_mode = (h->scored & need) == 0 ? rank_fast_mode(h, strat, sel_size, cap, nullptr) : 0;
    const bool want_auto = allow_auto && strat == SDPCUT_STRAT_COMB;
    // (the short-list predicate must be the one topk_select_enqueue will evaluate: it sees the tie-aware combined modes only when
    // the regime is resolved on the device -- a combined round whose mode the HOST resolved runs as TK_MODE_STRONG)
    const bool count_digit = h->fuse_keys && topk_fuse_ok(h, cap, want_auto);
    if (fast_mode && (want_auto || count_digit)) {
        void *ws = nullptr;
        rc = topk_begin(h, &ws, nullptr);
        if (rc) return rc;
        int64_t *strong = (need == (SDPCUT_EIG | SDPCUT_NN)) ? topk_strong_counter(ws) : nullptr;
        bool counted = false;
        if (count_digit) {
            ScoreFuse fuse;
            fuse.ws = ws;
            fuse.mode = fast_mode;
            fuse.k = (h->prefilter && h->N >= SDPCUT_PF_MIN_N) ? cap : 0;
            rc = launch_score(h, need, &fuse, &counted, strong);
        } else {
            rc = launch_score(h, need, nullptr, nullptr, strong);
        }
        if (rc) return rc;
        h->scored |= need;
        *stage = counted ? 3 : 1;
        *auto_out = want_auto;
    } else if ((h->scored & need) != need) {
        rc = sdpcut_score(h, need & ~h->scored);
        if (rc) return rc;
    }
    return 0;
}

extern "C" {

int sdpcut_destroy(sdpcut_handle h)
{
    if (!h) return SDPCUT_OK;
    hipSetDevice(h->device);
    sdpcut_sync(h);
    free_candidates(h);
    free_rank_ws(h);
    free_topk_ws(h);
    (void)hipFree(h->d_tri); (void)hipFree(h->d_tri_dense3);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->point_stage) (void)hipHostFree(h->point_stage);
    (void)hipFree(h->d_done_ticket);
    for (int k = 0; k <= SDPCUT_MAX_K; ++k) hipFree(h->net[k].d_blob);
    hipFree(h->d_Q); hipFree(h->d_vars); hipFree(h->d_counters); hipFree(h->d_stage); hipFree(h->d_stats);
    for (int i = 0; i < 4; ++i) if (h->ev[i]) hipEventDestroy(h->ev[i]);
    for (int i = 0; i < 3; ++i) {
        if (h->side_stream[i]) hipStreamDestroy(h->side_stream[i]);
        if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return SDPCUT_OK;
}

int sdpcut_set_option(sdpcut_handle h, int option, int64_t value)
{
    if (!h) return SDPCUT_EINVAL;
    switch (option) {
    case SDPCUT_OPT_KERNEL:
        if (value != SDPCUT_KERNEL_MFMA && value != SDPCUT_KERNEL_SIMPLE && value != SDPCUT_KERNEL_VALU)
            return sdpcut_fail(h, SDPCUT_EINVAL, "unknown kernel variant");
        h->kernel_variant = (int)value;
        return SDPCUT_OK;
    case SDPCUT_OPT_FUSE_KEYS:
        h->fuse_keys = value != 0;
        return SDPCUT_OK;
    case SDPCUT_OPT_COOP_LAUNCH:
        h->coop_launch = value != 0;
        return SDPCUT_OK;
    case SDPCUT_OPT_FUSED_TAIL:
        h->fused_tail = value != 0;
        return SDPCUT_OK;
    case SDPCUT_OPT_AUTO_REGIME:
        h->auto_regime = value != 0;
        return SDPCUT_OK;
    case SDPCUT_OPT_EIG_KERNEL:
        h->eig_kernel = value != 0;
        return SDPCUT_OK;
    case SDPCUT_OPT_TIMING:
        h->timing = value <= 0 ? 0 : (value == 1 ? 1 : 2);
        return SDPCUT_OK;
    case SDPCUT_OPT_ONE_LAUNCH:
        h->one_launch = value != 0;
        return SDPCUT_OK;
    case SDPCUT_OPT_PREFILTER:
        h->prefilter = value != 0;
        return SDPCUT_OK;
    case SDPCUT_OPT_SIDE_STREAMS:
        if (value < 0 || value > 2) return sdpcut_fail(h, SDPCUT_EINVAL, "SDPCUT_OPT_SIDE_STREAMS: 0 off, 1 on, 2 measured");
        h->side_streams = (int)value;
        h->side_choice = -1;
        return SDPCUT_OK;
    case SDPCUT_OPT_STREAM_PRIORITY: {
        SDPCUT_NO_PENDING(h);
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, sdpcut_sync(h));
        int least = 0, greatest = 0;
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
        hipStream_t s = nullptr;
        // (0 = the priority hipStreamCreateWithFlags gives: normal, which lies between the two ends of the range)
        HIP_TRY(h, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, value != 0 ? greatest : (0 < greatest ? greatest : (0 > least ? least : 0))));
        const bool on_own = h->stream == h->own_stream;
        if (h->own_stream) {
            HIP_TRY(h, hipStreamSynchronize(h->own_stream));
            (void)hipStreamDestroy(h->own_stream);
        }
        h->own_stream = s;
        if (on_own) h->stream = s;
        return SDPCUT_OK;
    }
    }
    return sdpcut_fail(h, SDPCUT_EINVAL, "unknown option");
}

int sdpcut_get_stat(sdpcut_handle h, int which, int64_t *value)
{
    if (!h || !value) return SDPCUT_EINVAL;
    switch (which) {
    case SDPCUT_STAT_ROUNDS: *value = h->stat_rounds; return SDPCUT_OK;
    case SDPCUT_STAT_SELECT_FALLBACKS: *value = h->stat_fallbacks; return SDPCUT_OK;
    case SDPCUT_STAT_SCORED: *value = h->have_point ? (int64_t)h->scored : 0; return SDPCUT_OK;
    case SDPCUT_STAT_TIE_SPLITS: *value = h->stat_tie_splits; return SDPCUT_OK;
    case SDPCUT_STAT_DIRECT_SELECTIONS:
    case SDPCUT_STAT_PF_BIN:
    case SDPCUT_STAT_PF_FLOOR:
    case SDPCUT_STAT_PF_COUNT: {
        unsigned long long v = 0;
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(&v, h->d_stats + (which - SDPCUT_STAT_DIRECT_SELECTIONS), sizeof(v), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, sdpcut_sync(h));
        *value = (int64_t)v;
        return SDPCUT_OK;
    }
    }
    return sdpcut_fail(h, SDPCUT_EINVAL, "unknown statistic");
}

int sdpcut_set_stream(sdpcut_handle h, void *hip_stream)
{
    if (!h) return SDPCUT_EINVAL;
    if (h->point_inflight) (void)sdpcut_sync(h);     // the staging copy's transfer lives on the old stream
    h->stream = (hip_stream == SDPCUT_OWN_STREAM) ? h->own_stream : (hipStream_t)hip_stream;
    h->topk_alt_clean = false;   // its zeroing was ordered on the previous stream only
    return SDPCUT_OK;
}

static __global__ void wake_kernel() {}

int sdpcut_wake(sdpcut_handle h)
{
    if (!h) return SDPCUT_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(wake_kernel, dim3(1), dim3(64), 0, h->stream);
    HIP_TRY(h, hipGetLastError());
    return SDPCUT_OK;
}

int sdpcut_synchronize(sdpcut_handle h)
{
    if (!h) return SDPCUT_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

int sdpcut_set_network(sdpcut_handle h, int k, int n_layers, const int32_t *widths, const double *params,
                       int64_t n_params)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (k < 2 || k > SDPCUT_MAX_K) return sdpcut_fail(h, SDPCUT_EINVAL, "k must be 2..5");
    if (n_layers < 2 || n_layers > MAX_LAYERS || !widths || !params)
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad layer description");
    const int d_in = k * (k + 3) / 2;
    const int nh = n_layers - 1;
    const int H = widths[0];
    if (widths[n_layers - 1] != 1) return sdpcut_fail(h, SDPCUT_EINVAL, "last layer must have one output");
    for (int l = 0; l < nh; ++l)
        if (widths[l] != H || H < 1 || H > MAX_HIDDEN)
            return sdpcut_fail(h, SDPCUT_EINVAL, "hidden layers must share one width <= 64");
    int64_t need = 2 * d_in + 1 + 3;
    {
        int fan = d_in;
        for (int l = 0; l < n_layers; ++l) { need += (int64_t)widths[l] * fan + widths[l]; fan = widths[l]; }
    }
    if (need != n_params) return sdpcut_fail(h, SDPCUT_EINVAL, "n_params does not match the layer description");
    HIP_TRY(h, hipSetDevice(h->device));

    // ---- unpack
    const double *p = params;
    const double *xoffset = p; p += d_in;
    const double *gain = p; p += d_in;
    const double ymin = *p++;
    const double *W[MAX_LAYERS], *B[MAX_LAYERS];
    {
        int fan = d_in;
        for (int l = 0; l < n_layers; ++l) {
            W[l] = p; p += (int64_t)widths[l] * fan;
            B[l] = p; p += widths[l];
            fan = widths[l];
        }
    }
    const double y_ymin = p[0], y_gain = p[1], y_xoffset = p[2];

    // ---- pack the device blob: inmap | bias | wout | raw W,b | A-fragments
    const int T = 4;
    const int s0 = (d_in + 3) / 4, sh = (H + 3) / 4;
    std::vector<double> blob;
    auto reserve = [&](size_t n) { size_t o = blob.size(); blob.resize(o + n, 0.0); return o; };
    const size_t o_inmap = reserve(2 * d_in);
    for (int i = 0; i < d_in; ++i) { blob[o_inmap + i] = xoffset[i]; blob[o_inmap + d_in + i] = gain[i]; }
    const size_t o_bias = reserve((size_t)nh * 64);
    for (int l = 0; l < nh; ++l)
        for (int j = 0; j < H; ++j) blob[o_bias + l * 64 + j] = B[l][j];
    const size_t o_bias_q = reserve((size_t)nh * 64);
    for (int l = 0; l < nh; ++l)
        for (int j = 0; j < H; ++j) blob[o_bias_q + l * 64 + j] = -0.25 * B[l][j];
    const size_t o_wout = reserve(64);
    for (int j = 0; j < H; ++j) blob[o_wout + j] = W[nh][j];
    size_t o_rw[MAX_LAYERS], o_rb[MAX_LAYERS];
    {
        int fan = d_in;
        for (int l = 0; l < n_layers; ++l) {
            o_rw[l] = reserve((size_t)widths[l] * fan);
            std::memcpy(&blob[o_rw[l]], W[l], sizeof(double) * widths[l] * fan);
            o_rb[l] = reserve(widths[l]);
            std::memcpy(&blob[o_rb[l]], B[l], sizeof(double) * widths[l]);
            fan = widths[l];
        }
    }
    // A-fragment of v_mfma_f64_16x16x4_f64: lane l holds A[row = l & 15][k = l >> 4]
    // => frag[t][s][l] = W[16 t + (l & 15)][4 s + (l >> 4)], zero outside the matrix
    const size_t o_frag = reserve((size_t)T * (s0 + (size_t)(nh - 1) * sh) * 64);
    {
        size_t o = o_frag;
        int fan = d_in;
        for (int l = 0; l < nh; ++l) {
            const int S = (l == 0) ? s0 : sh;
            for (int t = 0; t < T; ++t)
                for (int s = 0; s < S; ++s)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int row = 16 * t + (ln & 15), col = 4 * s + (ln >> 4);
                        // pre-scaled by -1/4 (exact: a power of two), see NetDev::bias_q
                        blob[o++] = (row < H && col < fan) ? -0.25 * W[l][(size_t)row * fan + col] : 0.0;
                    }
            fan = H;
        }
    }
    // rows 48..51 of every hidden layer, for the VALU tail of the MFMA kernel: [layer][4][64]
    const size_t o_wtail = reserve((size_t)nh * 4 * 64);
    {
        int fan = d_in;
        for (int l = 0; l < nh; ++l) {
            for (int u = 0; u < 4; ++u)
                for (int i = 0; i < fan; ++i)
                    if (48 + u < H) blob[o_wtail + ((size_t)l * 4 + u) * 64 + i] = -0.25 * W[l][(size_t)(48 + u) * fan + i];
            fan = H;
        }
    }
    // scalar-operand packing of the VALU kernel: [layer][j/8][i][j%8]
    const int NBv = (H + 7) / 8;
    // (+16: the kernel streams the weights in 16-double batches and may read past an odd fan-in)
    const size_t o_wvalu = reserve((size_t)NBv * 8 * ((size_t)d_in + (size_t)(nh - 1) * H) + 16);
    {
        size_t o = o_wvalu;
        int fan = d_in;
        for (int l = 0; l < nh; ++l) {
            for (int jb = 0; jb < NBv; ++jb)
                for (int i = 0; i < fan; ++i)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = jb * 8 + jj;
                        blob[o++] = (j < H) ? W[l][(size_t)j * fan + i] : 0.0;
                    }
            fan = H;
        }
    }
    NetHost &nh_ = h->net[k];
    HIP_TRY(h, sdpcut_sync(h));
    hipFree(nh_.d_blob);
    nh_.d_blob = nullptr;
    nh_.set = false;
    HIP_TRY(h, hipMalloc((void **)&nh_.d_blob, blob.size() * sizeof(double)));
    HIP_TRY(h, hipMemcpy(nh_.d_blob, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice));
    NetDev &d = nh_.dev;
    d = NetDev{};
    d.d_in = d_in; d.n_hidden = nh; d.width = H; d.s0 = s0; d.sh = sh;
    d.inmap = nh_.d_blob + o_inmap;
    d.bias = nh_.d_blob + o_bias;
    d.bias_q = nh_.d_blob + o_bias_q;
    d.wout = nh_.d_blob + o_wout;
    d.wfrag = nh_.d_blob + o_frag;
    d.wvalu = nh_.d_blob + o_wvalu;
    d.wtail = nh_.d_blob + o_wtail;
    for (int l = 0; l < n_layers; ++l) { d.raw_w[l] = nh_.d_blob + o_rw[l]; d.raw_b[l] = nh_.d_blob + o_rb[l]; }
    d.ymin = ymin; d.b_out = B[nh][0]; d.y_ymin = y_ymin; d.y_gain = y_gain; d.y_xoffset = y_xoffset;
    {
        // bound of every hidden pre-activation: |n_j| <= sum_i |W_ji| max|in_i| + |b_j| with |in| <= SDPCUT_INPUT_CLAMP
        // for the mapped inputs (x in [0,1], |q| <= 1/k map into [-1,1]) and <= 1 behind a tansig.  The tansig4 path
        // needs -2n <= 176, the tail rows -2n <= 704; 80 leaves a factor of two.
        double worst = 0.0;
        int fan = d_in;
        for (int l = 0; l < nh; ++l) {
            const double in_max = l == 0 ? SDPCUT_INPUT_CLAMP : 1.0;
            for (int j = 0; j < H; ++j) {
                double acc = std::fabs(B[l][j]);
                for (int i = 0; i < fan; ++i) acc += std::fabs(W[l][(size_t)j * fan + i]) * in_max;
                worst = acc > worst ? acc : worst;
            }
            fan = H;
        }
        d.unclamped_ok = worst < 40.0 ? 1 : 0;      // |n| < 40  <=>  -2n < 80
    }
    nh_.set = true;
    return SDPCUT_OK;
}

int sdpcut_set_instance(sdpcut_handle h, int32_t nb_vars, const double *Q_arr)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (nb_vars < 2 || nb_vars > 40000 || !Q_arr) return sdpcut_fail(h, SDPCUT_EINVAL, "bad instance");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, sdpcut_sync(h));
    const int64_t L = (int64_t)nb_vars * (nb_vars + 1) / 2;
    hipFree(h->d_Q); hipFree(h->d_vars);
    h->d_Q = nullptr; h->d_vars = nullptr; h->have_point = false; h->scored = 0;
    HIP_TRY(h, hipMalloc((void **)&h->d_Q, L * sizeof(double)));
    HIP_TRY(h, hipMalloc((void **)&h->d_vars, (L + nb_vars) * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->d_Q, Q_arr, L * sizeof(double), hipMemcpyHostToDevice));
    h->nb_vars = nb_vars;
    h->L = L;
    return SDPCUT_OK;
}

int sdpcut_set_candidates(sdpcut_handle h, int64_t N, const int32_t *set_inds, int32_t ld, const int32_t *ks,
                          int64_t global_base)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (h->nb_vars == 0) return sdpcut_fail(h, SDPCUT_ESTATE, "set_instance first");
    if (N < 0 || N > 0x7fffffffLL || (N > 0 && (!set_inds || !ks)) || ld < 2)
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad candidate list");
    HIP_TRY(h, hipSetDevice(h->device));
    // validate + bucket by size on the host (once per instance)
    int64_t cnt[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = 0; i < N; ++i) {
        const int k = ks[i];
        if (k < 2 || k > SDPCUT_MAX_K || k > ld) return sdpcut_fail(h, SDPCUT_EINVAL, "candidate size must be 2..5");
        for (int a = 0; a < k; ++a) {
            const int32_t v = set_inds[i * ld + a];
            if (v < 0 || v >= h->nb_vars) return sdpcut_fail(h, SDPCUT_EINVAL, "variable index out of range");
        }
        ++cnt[k];
    }
    std::vector<int32_t> pad((size_t)N * 5, -1), kk((size_t)N);
    std::vector<int32_t> soa[SDPCUT_MAX_K + 1], orig[SDPCUT_MAX_K + 1];
    int64_t fill[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
    for (int k = 2; k <= SDPCUT_MAX_K; ++k) { soa[k].resize((size_t)cnt[k] * k); orig[k].resize((size_t)cnt[k]); }
    for (int64_t i = 0; i < N; ++i) {
        const int k = ks[i];
        kk[i] = k;
        const int64_t p = fill[k]++;
        orig[k][p] = (int32_t)i;
        for (int a = 0; a < k; ++a) {
            const int32_t v = set_inds[i * ld + a];
            pad[i * 5 + a] = v;
            soa[k][(size_t)a * cnt[k] + p] = v;
        }
    }
    int rc = alloc_candidates(h, N, cnt, global_base);
    if (rc) return rc;
    if (N > 0) {
        HIP_TRY(h, hipMemcpy(h->d_set_orig, pad.data(), (size_t)N * 5 * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_k, kk.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    for (int k = 2; k <= SDPCUT_MAX_K; ++k) {
        Bucket &b = h->bucket[k];
        if (!cnt[k]) continue;
        HIP_TRY(h, hipMemcpy(b.d_set, soa[k].data(), soa[k].size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(b.d_orig, orig[k].data(), orig[k].size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return SDPCUT_OK;
}

static int ensure_point_stage(sdpcut_ctx *h)
{
    const size_t bytes = (size_t)(h->L + h->nb_vars) * sizeof(double);
    if (h->point_stage_bytes >= bytes) return 0;
    HIP_TRY(h, sdpcut_sync(h));
    if (h->point_stage) (void)hipHostFree(h->point_stage);
    h->point_stage = nullptr;
    h->point_stage_bytes = 0;
    HIP_TRY(h, hipHostMalloc(&h->point_stage, bytes, hipHostMallocMapped));
    HIP_TRY(h, hipHostGetDevicePointer(&h->point_stage_dev, h->point_stage, 0));
    h->point_stage_bytes = bytes;
    return 0;
}

int sdpcut_point_buffer(sdpcut_handle h, double **buf)
{
    if (!h) return SDPCUT_EINVAL;
    if (!buf) return sdpcut_fail(h, SDPCUT_EINVAL, "buf is NULL");
    if (!h->d_vars) return sdpcut_fail(h, SDPCUT_ESTATE, "set_instance first");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_point_stage(h);
    if (rc) return rc;
    *buf = (double *)h->point_stage;
    return SDPCUT_OK;
}

int sdpcut_set_point(sdpcut_handle h, const double *vars_values)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (!h->d_vars) return sdpcut_fail(h, SDPCUT_ESTATE, "set_instance first");
    if (!vars_values) return sdpcut_fail(h, SDPCUT_EINVAL, "vars_values is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    // The caller's (pageable) buffer is copied into a pinned staging block and sent from there: the
    // call returns as soon as the host copy is done -- the caller may reuse its buffer at once -- and
    // the DMA runs behind it on the stream, in front of the score kernels (no blocking round trip
    // per round).  The staging block is reused once the previous transfer out of it has completed.
    // A caller that wrote the point straight into the staging block (sdpcut_point_buffer) skips the copy.
    const size_t bytes = (size_t)(h->L + h->nb_vars) * sizeof(double);
    int rc = ensure_point_stage(h);
    if (rc) return rc;
    if (vars_values != (const double *)h->point_stage) {
        // (every round ends in a host wait on the device, so this one is normally skipped; an event per
        // transfer would put a barrier packet -- ~10 us -- in front of every score launch)
        if (h->point_inflight) HIP_TRY(h, sdpcut_sync(h));
        std::memcpy(h->point_stage, vars_values, bytes);
    }
    // a kernel of the compute queue pulls the block over PCIe (mapped host memory): the score launch
    // follows it in queue order, whereas a copy-engine transfer costs a cross-queue hand-off (~10 us)
    // in front of every round
    rc = launch_point_copy(h, (const double *)h->point_stage_dev, h->L + h->nb_vars);
    if (rc) return rc;
    h->point_inflight = true;
    h->have_point = true;
    h->scored = 0;
    h->last_total = -1;
    return SDPCUT_OK;
}

int sdpcut_set_point_device(sdpcut_handle h, const void *d_vars_values)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (!h->d_vars) return sdpcut_fail(h, SDPCUT_ESTATE, "set_instance first");
    if (!d_vars_values) return sdpcut_fail(h, SDPCUT_EINVAL, "d_vars_values is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->d_vars, d_vars_values, (h->L + h->nb_vars) * sizeof(double),
                              hipMemcpyDeviceToDevice, h->stream));
    h->have_point = true;
    h->scored = 0;
    h->last_total = -1;
    return SDPCUT_OK;
}

int sdpcut_score(sdpcut_handle h, uint32_t flags)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (!(flags & (SDPCUT_EIG | SDPCUT_NN)) || (flags & ~(uint32_t)(SDPCUT_EIG | SDPCUT_NN)))
        return sdpcut_fail(h, SDPCUT_EINVAL, "flags must be a combination of SDPCUT_EIG and SDPCUT_NN");
    if (!h->have_point) return sdpcut_fail(h, SDPCUT_ESTATE, "set_point first");
    if (!h->d_eig) return sdpcut_fail(h, SDPCUT_ESTATE, "set_candidates first");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = launch_score(h, flags);   // with SDPCUT_OPT_TIMING the dispatches carry ev[0] / ev[1]
    if (rc) return rc;
    h->scored |= flags;
    return SDPCUT_OK;
}

int sdpcut_get_scores(sdpcut_handle h, double *eigmin, double *obj_improve)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (eigmin) {
        if (!(h->scored & SDPCUT_EIG)) return sdpcut_fail(h, SDPCUT_ESTATE, "eigenvalues not scored");
        HIP_TRY(h, hipMemcpyAsync(eigmin, h->d_eig, h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    if (obj_improve) {
        if (!(h->scored & SDPCUT_NN)) return sdpcut_fail(h, SDPCUT_ESTATE, "optimality measure not scored");
        HIP_TRY(h, hipMemcpyAsync(obj_improve, h->d_obj, h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

static int check_rank_args(sdpcut_ctx *h, int strat)
{
    const bool part = strat == SDPCUT_PART_STRONG;
    if (strat != SDPCUT_STRAT_FEAS && strat != SDPCUT_STRAT_OPT && strat != SDPCUT_STRAT_COMB && !part)
        return sdpcut_fail(h, SDPCUT_EINVAL, "strategy must be 1 (feasibility), 2 (optimality) or 4 (combined)");
    const uint32_t need = strat == SDPCUT_STRAT_FEAS ? SDPCUT_EIG
                          : strat == SDPCUT_STRAT_OPT ? SDPCUT_NN : (SDPCUT_EIG | SDPCUT_NN);
    if ((h->scored & need) != need) return sdpcut_fail(h, SDPCUT_ESTATE, "sdpcut_score with the needed flags first");
    return 0;
}

int sdpcut_rank_device(sdpcut_handle h, int strat, int64_t sel_size, int64_t max_out, void *d_idx_out,
                       void *d_score_out, int64_t *n_written, int64_t *n_total, int32_t *new_strat,
                       int64_t *counters)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    int rc = check_rank_args(h, strat);
    if (rc) return rc;
    if (max_out < 0 || (max_out > 0 && (!d_idx_out || !d_score_out))) return sdpcut_fail(h, SDPCUT_EINVAL, "bad output");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->timing > 1) HIP_TRY(h, hipEventRecord(h->ev[2], h->stream));
    rc = rank_on_device(h, strat, sel_size, max_out, (int64_t *)d_idx_out, (double *)d_score_out, n_written, n_total,
                        new_strat, counters);
    if (rc) return rc;
    if (h->timing > 1) HIP_TRY(h, hipEventRecord(h->ev[3], h->stream));
    return SDPCUT_OK;
}

int sdpcut_rank(sdpcut_handle h, int strat, int64_t sel_size, int64_t max_out, int64_t *idx_out, double *score_out,
                int64_t *n_total, int32_t *new_strat, int64_t *counters)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (max_out < 0 || (max_out > 0 && (!idx_out || !score_out))) return sdpcut_fail(h, SDPCUT_EINVAL, "bad output");
    HIP_TRY(h, hipSetDevice(h->device));
    int64_t cap = max_out < h->N ? max_out : h->N;
    int rc = ensure_stage(h, (size_t)(cap < 1 ? 1 : cap) * 16);
    if (rc) return rc;
    int64_t *d_idx = (int64_t *)h->d_stage;
    double *d_sc = (double *)((char *)h->d_stage + (size_t)(cap < 1 ? 1 : cap) * 8);
    int64_t w = 0;
    rc = sdpcut_rank_device(h, strat, sel_size, cap, d_idx, d_sc, &w, n_total, new_strat, counters);
    if (rc) return rc;
    if (w > 0) {
        HIP_TRY(h, hipMemcpyAsync(idx_out, d_idx, w * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(score_out, d_sc, w * 8, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

int sdpcut_rank_fetch(sdpcut_handle h, int64_t offset, int64_t count, int64_t *idx_out, double *score_out)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (h->last_total < 0) return sdpcut_fail(h, SDPCUT_ESTATE, "no ranking available: call sdpcut_rank first");
    if (offset < 0 || count < 0 || offset + count > h->last_total || (count > 0 && (!idx_out || !score_out)))
        return sdpcut_fail(h, SDPCUT_EINVAL, "window outside the last ranking");
    if (count == 0) return SDPCUT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_stage(h, (size_t)count * 16);
    if (rc) return rc;
    int64_t *d_idx = (int64_t *)h->d_stage;
    double *d_sc = (double *)((char *)h->d_stage + (size_t)count * 8);
    rc = rank_fetch_on_device(h, offset, count, d_idx, d_sc);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(idx_out, d_idx, count * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(score_out, d_sc, count * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

int sdpcut_merge_topk_device(sdpcut_handle h, int64_t count, const void *d_scores, const void *d_secondary,
                             const void *d_ids, int64_t max_out, void *d_score_out, void *d_id_out)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (count < 0 || max_out < 0 || (count > 0 && max_out > 0 && (!d_scores || !d_ids || !d_score_out || !d_id_out)))
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad merge arguments");
    if (count > 0x7fffffffLL) return sdpcut_fail(h, SDPCUT_EINVAL, "merge too large");
    HIP_TRY(h, hipSetDevice(h->device));
    return merge_topk_on_device(h, count, (const double *)d_scores, (const double *)d_secondary,
                                (const int64_t *)d_ids, max_out, (double *)d_score_out, (int64_t *)d_id_out);
}

int sdpcut_gather_scores_device(sdpcut_handle h, int64_t count, const void *d_ids, void *d_eig_out, void *d_obj_out)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (count < 0 || (count > 0 && !d_ids)) return sdpcut_fail(h, SDPCUT_EINVAL, "bad gather arguments");
    if ((d_eig_out && !(h->scored & SDPCUT_EIG)) || (d_obj_out && !(h->scored & SDPCUT_NN)))
        return sdpcut_fail(h, SDPCUT_ESTATE, "sdpcut_score with the needed flags first");
    HIP_TRY(h, hipSetDevice(h->device));
    return gather_scores_on_device(h, count, (const int64_t *)d_ids, (double *)d_eig_out, (double *)d_obj_out);
}

int sdpcut_cut_rows(sdpcut_handle h, int64_t count, const int64_t *idx, double *lam_min, double *coef, double *rhs,
                    int64_t *cols, int32_t *ks)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (!h->have_point || !h->d_set_orig) return sdpcut_fail(h, SDPCUT_ESTATE, "set_candidates and set_point first");
    if (count < 0 || (count > 0 && (!idx || !lam_min || !coef || !rhs || !cols || !ks)))
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad cut_rows arguments");
    if (count == 0) return SDPCUT_OK;
    for (int64_t i = 0; i < count; ++i)
        if (idx[i] < 0 || idx[i] >= h->N) return sdpcut_fail(h, SDPCUT_EINVAL, "candidate index out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    // staging layout: idx | lam | rhs | coef | cols | ks
    const size_t c = (size_t)count;
    const size_t bytes = c * 8 * (3 + 2 * SDPCUT_ROW_LD) + c * 4;
    int rc = ensure_stage(h, bytes);
    if (rc) return rc;
    char *p = (char *)h->d_stage;
    int64_t *d_idx = (int64_t *)p; p += c * 8;
    double *d_lam = (double *)p; p += c * 8;
    double *d_rhs = (double *)p; p += c * 8;
    double *d_coef = (double *)p; p += c * 8 * SDPCUT_ROW_LD;
    int64_t *d_cols = (int64_t *)p; p += c * 8 * SDPCUT_ROW_LD;
    int32_t *d_ks = (int32_t *)p;
    HIP_TRY(h, hipMemcpyAsync(d_idx, idx, c * 8, hipMemcpyHostToDevice, h->stream));
    rc = launch_cut_rows(h, count, nullptr, d_idx, 0, d_lam, d_coef, SDPCUT_ROW_LD, d_rhs, d_cols, d_ks);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(lam_min, d_lam, c * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(rhs, d_rhs, c * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(coef, d_coef, c * 8 * SDPCUT_ROW_LD, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(cols, d_cols, c * 8 * SDPCUT_ROW_LD, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(ks, d_ks, c * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

} // extern "C"

// One fused round in two halves: round_begin enqueues everything -- score (if needed) -> rank -> epilogue into the pinned block --
// WITHOUT waiting; round_end waits, reads the counters and, when the enqueued selection is not the answer (a void one; the
// combined scan of heads > 8192), runs the general path.  csr = false: padded rows, block layout of sdpcut_select_round_view;
// csr = true: the CSR block of sdpcut_round_csr (rows.hip, csr_layout).  Two handles may both begin before either ends: their
// device work overlaps (the QCQP round's two covers).
static int round_begin(sdpcut_ctx *h, int strat, int64_t sel_size, int32_t coef_ld, bool csr)
{
    if (strat != SDPCUT_STRAT_FEAS && strat != SDPCUT_STRAT_OPT && strat != SDPCUT_STRAT_COMB)
        return sdpcut_fail(h, SDPCUT_EINVAL, "strategy must be 1 (feasibility), 2 (optimality) or 4 (combined)");
    if (sel_size < 0) return sdpcut_fail(h, SDPCUT_EINVAL, "bad select_round arguments");
    if (!h->have_point || !h->d_eig) return sdpcut_fail(h, SDPCUT_ESTATE, "set_candidates and set_point first");
    if (coef_ld < h->row_len_max || coef_ld > SDPCUT_ROW_LD)
        return sdpcut_fail(h, SDPCUT_EINVAL, "coef_ld must hold the longest row (k + k(k+1)/2) and be <= SDPCUT_ROW_LD");
    SDPCUT_NO_PENDING(h);        // a fused round begun, or a sharded round enqueued and not waited for
    HIP_TRY(h, hipSetDevice(h->device));
    const uint32_t need = strat == SDPCUT_STRAT_FEAS ? SDPCUT_EIG
                          : strat == SDPCUT_STRAT_OPT ? SDPCUT_NN : (SDPCUT_EIG | SDPCUT_NN);
    int rc;
    int64_t cap = sel_size < h->N ? sel_size : h->N;
    int stage = 0;               // how far the selection's first pass has got (topk_select_enqueue)
    bool auto_regime = false;
    rc = score_for_selection(h, strat, sel_size, cap, need, h->auto_regime, &stage, &auto_regime);
    if (rc) return rc;
    PendingRound &P = h->pend;
    P = PendingRound();
    P.strat = strat; P.sel_size = sel_size; P.cap = cap; P.ld = coef_ld; P.csr = csr;
    if (cap == 0) {   // nothing to generate; round_end still reports the ranking's length / strategy switch
        P.active = true;
        return SDPCUT_OK;
    }
    // one block for everything a round returns: counters | idx | score | lam | rhs | coef | ks
    const size_t c = (size_t)cap;
    const size_t ret_bytes = csr ? csr_layout(cap, coef_ld).bytes : 64 + c * 8 * (4 + (size_t)coef_ld) + c * 4;
    rc = ensure_stage(h, 64 + c * 8 * (4 + (size_t)coef_ld) + c * 4 + 64);
    if (rc) return rc;
    rc = ensure_pinned(h, ret_bytes);
    if (rc) return rc;
    int64_t *d_idx = (int64_t *)((char *)h->d_stage + 64);
    double *d_sc = (double *)(d_idx + c);
    if (h->timing > 1) HIP_TRY(h, hipEventRecord(h->ev[2], h->stream));
    int64_t *hdr = (int64_t *)h->pinned;
    if (csr) hdr[8] = hdr[9] = hdr[10] = 0;
    // fast path: selection and rows are enqueued back to back; the epilogue kernel stores the
    // results directly into the pinned host block (no copy engine); one synchronisation
    const int64_t *d_cnt = nullptr;
    rc = rank_fast_enqueue(h, strat, sel_size, cap, d_idx, d_sc, &d_cnt, stage, auto_regime);
    if (rc < 0) return rc;
    P.fast_tried = rc == 1;
    if (P.fast_tried) {
        if (h->timing > 1) HIP_TRY(h, hipEventRecord(h->ev[3], h->stream));
        // the epilogue's last workgroup publishes this round's serial number in the block's header: the
        // host polls that word instead of waiting for the runtime's completion signal (~5 us earlier)
        P.serial = ++h->round_serial;
        rc = csr ? launch_round_csr(h, cap, d_cnt, cap, d_idx, d_sc, coef_ld, h->pinned_dev, P.serial)
                 : launch_round_rows(h, cap, d_cnt, d_idx, d_sc, coef_ld, h->pinned_dev, 64, P.serial);
        if (rc) return rc;
    }
    P.active = true;
    return SDPCUT_OK;
}

// round_csr_kernel's look-back over lower-indexed workgroups is bounded (CSR_SPIN_LIMIT): on a device shared with a kernel that
// blocks it for long it gives up and sets hdr[10].  That is a transient condition, not an error of the round: by the time the
// host sees the mark every workgroup of that launch has retired, so ONE more launch over the same head (ids and scores still
// in the staging area) finds its predecessors' aggregates as soon as they are dispatched.  Counted like the selection's
// fallbacks; only a second give-up fails the call.
static int csr_again(sdpcut_ctx *h, int64_t cap, int64_t w, const int64_t *d_idx, const double *d_sc, int32_t coef_ld)
{
    int64_t *hdr = (int64_t *)h->pinned;
    hdr[8] = hdr[9] = hdr[10] = 0;
    ++h->stat_fallbacks;
    const int64_t serial = ++h->round_serial;
    int rc = launch_round_csr(h, cap, nullptr, w, d_idx, d_sc, coef_ld, h->pinned_dev, serial);
    if (rc) return rc;
    rc = wait_round_done(h, hdr + 7, serial);
    if (rc) return rc;
    if (hdr[10]) return sdpcut_fail(h, SDPCUT_EHIP, "round_csr: look-back of the row assembly timed out twice");
    return SDPCUT_OK;
}

static int round_end(sdpcut_ctx *h, const void **block, int64_t *cap_out, int64_t *n_out, int64_t *n_total, int32_t *new_strat,
                     int64_t *counters)
{
    if (!block || !cap_out || !n_out) return sdpcut_fail(h, SDPCUT_EINVAL, "bad select_round arguments");
    if (!h->pend.active) return sdpcut_fail(h, SDPCUT_ESTATE, "no round pending on this handle");
    const PendingRound P = h->pend;
    h->pend.active = false;
    HIP_TRY(h, hipSetDevice(h->device));
    const int strat = P.strat;
    const int64_t sel_size = P.sel_size, cap = P.cap;
    const int32_t coef_ld = P.ld;
    const bool csr = P.csr;
    *n_out = 0;
    *cap_out = cap;
    *block = nullptr;
    if (cap == 0) return sdpcut_rank(h, strat, sel_size, 0, nullptr, nullptr, n_total, new_strat, counters);
    const size_t c = (size_t)cap;
    char *p = (char *)h->d_stage;
    p += 64;
    int64_t *d_idx = (int64_t *)p; p += c * 8;
    double *d_sc = (double *)p; p += c * 8;
    double *d_lam = (double *)p; p += c * 8;
    double *d_rhs = (double *)p; p += c * 8;
    double *d_coef = (double *)p; p += c * 8 * (size_t)coef_ld;
    int32_t *d_ks = (int32_t *)p;
    int64_t *hdr = (int64_t *)h->pinned;
    int rc;
    int64_t w = 0;
    bool have = false;
    if (P.fast_tried) {
        rc = wait_round_done(h, hdr + 7, P.serial);
        if (rc) return rc;
        have = rank_fast_finish(h, strat, sel_size, cap, (const int64_t *)h->pinned, &w, n_total, new_strat, counters) != 0;
        if (have && csr && hdr[10] && w > 0) {      // the row assembly gave up its look-back (rows.hip): once more, see csr_again
            rc = csr_again(h, cap, w, d_idx, d_sc, coef_ld);
            if (rc) return rc;
        }
    }
    ++h->stat_rounds;
    bool resorted = false;      // the head in d_idx / d_sc was produced after the enqueued epilogue ran: launch it again
    if (P.fast_tried && !have && strat == SDPCUT_STRAT_COMB && hdr[4] == 2 && hdr[6] == 4 /* TK_MODE_COMBALL */) {
        // the threshold tie group of the every-entry-visited ranking does not fit the sort buffers (a structured LP vertex):
        // cut it by its secondary key with two more selections (topk.hip: topk_tie_split) instead of sorting the full list
        int64_t c7[7];
        for (int i = 0; i < 7; ++i) c7[i] = hdr[i];
        rc = topk_tie_split(h, cap, d_idx, d_sc, nullptr);
        if (rc < 0) return rc;
        if (rc == 0) {
            c7[4] = 0;
            have = rank_fast_finish(h, strat, sel_size, cap, c7, &w, n_total, new_strat, counters) != 0;
            resorted = have;
        }
    }
    if (P.fast_tried && !have && hdr[4]) ++h->stat_fallbacks;
    if (resorted) {
        if (h->timing > 1) HIP_TRY(h, hipEventRecord(h->ev[3], h->stream));
        if (csr) {
            hdr[8] = hdr[9] = hdr[10] = 0;
            const int64_t serial = ++h->round_serial;
            rc = launch_round_csr(h, cap, nullptr, w, d_idx, d_sc, coef_ld, h->pinned_dev, serial);
            if (rc) return rc;
            rc = wait_round_done(h, hdr + 7, serial);
            if (rc) return rc;
            if (hdr[10] && (rc = csr_again(h, cap, w, d_idx, d_sc, coef_ld))) return rc;
        } else if (w > 0) {
            rc = launch_cut_rows(h, w, nullptr, d_idx, h->base, d_lam, d_coef, coef_ld, d_rhs, nullptr, d_ks);
            if (rc) return rc;
            HIP_TRY(h, hipMemcpyAsync(h->pinned, h->d_stage, 64 + c * 8 * (4 + (size_t)coef_ld) + c * 4, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, sdpcut_sync(h));
        }
    }
    if (!have) {
        // general path (full sorts; the combined scan visiting every entry, or heads > 8192)
        // (the fast attempt above already counted the strong candidates: no second attempt)
        const int64_t *c5 = (const int64_t *)h->pinned;
        const int64_t hint = (P.fast_tried && strat == SDPCUT_STRAT_COMB && !c5[4]) ? c5[0] : -1;
        rc = rank_on_device(h, strat, sel_size, cap, d_idx, d_sc, &w, n_total, new_strat, counters, hint);
        if (rc) return rc;
        if (h->timing > 1) HIP_TRY(h, hipEventRecord(h->ev[3], h->stream));
        if (w > 0 && csr) {
            hdr[8] = hdr[9] = hdr[10] = 0;
            const int64_t serial = ++h->round_serial;
            rc = launch_round_csr(h, cap, nullptr, w, d_idx, d_sc, coef_ld, h->pinned_dev, serial);
            if (rc) return rc;
            rc = wait_round_done(h, hdr + 7, serial);
            if (rc) return rc;
            if (hdr[10] && (rc = csr_again(h, cap, w, d_idx, d_sc, coef_ld))) return rc;
        } else if (w > 0) {
            rc = launch_cut_rows(h, w, nullptr, d_idx, h->base, d_lam, d_coef, coef_ld, d_rhs, nullptr, d_ks);
            if (rc) return rc;
            HIP_TRY(h, hipMemcpyAsync(h->pinned, h->d_stage, 64 + c * 8 * (4 + (size_t)coef_ld) + c * 4, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, sdpcut_sync(h));
        }
    }
    *n_out = w;
    *block = h->pinned;
    return SDPCUT_OK;
}

static int round_impl(sdpcut_ctx *h, int strat, int64_t sel_size, int32_t coef_ld, bool csr, const void **block,
                      int64_t *cap_out, int64_t *n_out, int64_t *n_total, int32_t *new_strat, int64_t *counters)
{
    if (!block || !cap_out || !n_out) return sdpcut_fail(h, SDPCUT_EINVAL, "bad select_round arguments");
    int rc = round_begin(h, strat, sel_size, coef_ld, csr);
    if (rc) return rc;
    return round_end(h, block, cap_out, n_out, n_total, new_strat, counters);
}

static int fill_csr_out(sdpcut_ctx *h, const void *block, sdpcut_round_csr_t *out)
{
    if (!block || out->cap == 0) return SDPCUT_OK;
    const CsrLayout y = csr_layout(out->cap, out->row_ld);
    const char *b = (const char *)block;
    const int64_t *hdr = (const int64_t *)b;
    out->idx = (const int64_t *)(b + y.idx);
    out->score = (const double *)(b + y.score);
    out->lam_min = (const double *)(b + y.lam);
    out->ks = (const int32_t *)(b + y.ks);
    out->set_inds = (const int32_t *)(b + y.sets);
    out->n_rows = out->n_out > 0 ? hdr[8] : 0;
    out->nnz = out->n_out > 0 ? hdr[9] : 0;
    out->row_entry = (const int32_t *)(b + y.row_entry);
    out->indptr = (const int32_t *)(b + y.indptr);
    out->indices = (const int32_t *)(b + y.indices);
    out->values = (const double *)(b + y.values);
    out->rhs = (const double *)(b + y.rhs);
    return SDPCUT_OK;
}

extern "C" {

int sdpcut_select_round_view(sdpcut_handle h, int strat, int64_t sel_size, int32_t coef_ld, const void **block,
                             int64_t *cap_out, int64_t *n_out, int64_t *n_total, int32_t *new_strat, int64_t *counters)
{
    if (!h) return SDPCUT_EINVAL;
    return round_impl(h, strat, sel_size, coef_ld, false, block, cap_out, n_out, n_total, new_strat, counters);
}

int sdpcut_round_csr_begin(sdpcut_handle h, const double *vars_values, int strat, int64_t sel_size)
{
    if (!h) return SDPCUT_EINVAL;
    int rc;
    if (vars_values && (rc = sdpcut_set_point(h, vars_values))) return rc;
    return round_begin(h, strat, sel_size, h->row_len_max, true);
}

int sdpcut_round_csr_end(sdpcut_handle h, sdpcut_round_csr_t *out)
{
    if (!h) return SDPCUT_EINVAL;
    if (!out) return sdpcut_fail(h, SDPCUT_EINVAL, "out is NULL");
    std::memset(out, 0, sizeof(*out));
    if (!h->pend.active || !h->pend.csr) return sdpcut_fail(h, SDPCUT_ESTATE, "no sdpcut_round_csr_begin pending on this handle");
    const void *block = nullptr;
    out->row_ld = h->pend.ld;
    int rc = round_end(h, &block, &out->cap, &out->n_out, &out->n_total, &out->new_strat, out->counters);
    if (rc) return rc;
    return fill_csr_out(h, block, out);
}

int sdpcut_round_csr(sdpcut_handle h, const double *vars_values, int strat, int64_t sel_size, sdpcut_round_csr_t *out)
{
    if (!h) return SDPCUT_EINVAL;
    if (!out) return sdpcut_fail(h, SDPCUT_EINVAL, "out is NULL");
    int rc = sdpcut_round_csr_begin(h, vars_values, strat, sel_size);
    if (rc) { std::memset(out, 0, sizeof(*out)); return rc; }
    return sdpcut_round_csr_end(h, out);
}

int sdpcut_round_view(sdpcut_handle h, const double *vars_values, int strat, int64_t sel_size, int32_t coef_ld,
                      const void **block, int64_t *cap_out, int64_t *n_out, int64_t *n_total, int32_t *new_strat,
                      int64_t *counters)
{
    int rc = sdpcut_set_point(h, vars_values);
    if (rc) return rc;
    return sdpcut_select_round_view(h, strat, sel_size, coef_ld, block, cap_out, n_out, n_total, new_strat, counters);
}

int sdpcut_select_round(sdpcut_handle h, int strat, int64_t sel_size, int32_t coef_ld, int64_t *idx_out,
                        double *score_out, double *lam_min, double *coef, double *rhs, int32_t *ks, int64_t *n_out,
                        int64_t *n_total, int32_t *new_strat, int64_t *counters)
{
    if (!h) return SDPCUT_EINVAL;
    if (sel_size < 0 || (sel_size > 0 && (!idx_out || !score_out || !lam_min || !coef || !rhs || !ks)) || !n_out)
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad select_round arguments");
    const void *block = nullptr;
    int64_t cap = 0;
    int rc = sdpcut_select_round_view(h, strat, sel_size, coef_ld, &block, &cap, n_out, n_total, new_strat, counters);
    if (rc) return rc;
    if (*n_out > 0) {
        const size_t c = (size_t)cap, ww = (size_t)*n_out;
        const char *q = (const char *)block + 64;
        std::memcpy(idx_out, q, ww * 8); q += c * 8;
        std::memcpy(score_out, q, ww * 8); q += c * 8;
        std::memcpy(lam_min, q, ww * 8); q += c * 8;
        std::memcpy(rhs, q, ww * 8); q += c * 8;
        std::memcpy(coef, q, ww * 8 * (size_t)coef_ld); q += c * 8 * (size_t)coef_ld;
        std::memcpy(ks, q, ww * 4);
    }
    return SDPCUT_OK;
}

int sdpcut_eig_batch(sdpcut_handle h, int k, int64_t count, const double *x_rho, const double *X_rho,
                     double *eigvals, double *evecs)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (k < 2 || k > SDPCUT_MAX_K) return sdpcut_fail(h, SDPCUT_EINVAL, "k must be 2..5");
    if (count < 0 || (count > 0 && (!x_rho || !X_rho || !eigvals))) return sdpcut_fail(h, SDPCUT_EINVAL, "bad eig_batch arguments");
    if (count == 0) return SDPCUT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t c = (size_t)count, m = (size_t)k * (k + 1) / 2, D = (size_t)k + 1;
    const size_t bytes = c * 8 * (k + m + D + D * D);
    int rc = ensure_stage(h, bytes);
    if (rc) return rc;
    double *d_x = (double *)h->d_stage, *d_X = d_x + c * k, *d_w = d_X + c * m, *d_v = d_w + c * D;
    HIP_TRY(h, hipMemcpyAsync(d_x, x_rho, c * k * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_X, X_rho, c * m * 8, hipMemcpyHostToDevice, h->stream));
    rc = launch_eig_batch(h, k, count, d_x, d_X, d_w, evecs ? d_v : nullptr);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(eigvals, d_w, c * D * 8, hipMemcpyDeviceToHost, h->stream));
    if (evecs) HIP_TRY(h, hipMemcpyAsync(evecs, d_v, c * D * D * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

int sdpcut_nn_batch(sdpcut_handle h, int k, int64_t count, const double *inputs, double *out)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (k < 2 || k > SDPCUT_MAX_K) return sdpcut_fail(h, SDPCUT_EINVAL, "k must be 2..5");
    if (!h->net[k].set) return sdpcut_fail(h, SDPCUT_ESTATE, "no network set for this candidate size");
    if (count < 0 || (count > 0 && (!inputs || !out))) return sdpcut_fail(h, SDPCUT_EINVAL, "bad nn_batch arguments");
    if (count == 0) return SDPCUT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t c = (size_t)count, d = (size_t)k * (k + 3) / 2;
    int rc = ensure_stage(h, c * 8 * (d + 1));
    if (rc) return rc;
    double *d_in = (double *)h->d_stage, *d_out = d_in + c * d;
    HIP_TRY(h, hipMemcpyAsync(d_in, inputs, c * d * 8, hipMemcpyHostToDevice, h->stream));
    rc = launch_nn_batch(h, k, count, d_in, d_out);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(out, d_out, c * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

int sdpcut_tri_preprocess(sdpcut_handle h, const uint8_t *adjacency, int64_t *n_triples)
{
    if (!h) return SDPCUT_EINVAL;
    if (h->nb_vars == 0) return sdpcut_fail(h, SDPCUT_ESTATE, "set_instance first");
    if (!adjacency) return sdpcut_fail(h, SDPCUT_EINVAL, "adjacency is NULL");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, sdpcut_sync(h));
    return tri_preprocess(h, adjacency, n_triples);
}

int sdpcut_tri_get_triples(sdpcut_handle h, int32_t *triples_out, uint8_t *density_out)
{
    if (!h) return SDPCUT_EINVAL;
    if (h->n_tri > 0 && !triples_out) return sdpcut_fail(h, SDPCUT_EINVAL, "triples_out is NULL");
    if (h->n_tri > 0) std::memcpy(triples_out, h->tri_host.data(), (size_t)h->n_tri * 3 * sizeof(int32_t));
    if (density_out)
        for (int64_t t = 0; t < h->n_tri; ++t) density_out[t] = h->tri_dense_host[t] ? 3 : 2;
    return SDPCUT_OK;
}

int sdpcut_tri_separate(sdpcut_handle h, int64_t max_out, int64_t *entry_out, double *viol_out, int64_t *n_violated,
                        int64_t *n_written)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (!h->have_point) return sdpcut_fail(h, SDPCUT_ESTATE, "set_point first");
    if (max_out < 0 || (max_out > 0 && (!entry_out || !viol_out)) || !n_violated || !n_written)
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad tri_separate arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    int64_t cap = max_out < 4 * h->n_tri ? max_out : 4 * h->n_tri;
    int rc = ensure_stage(h, (size_t)(cap < 1 ? 1 : cap) * 16);
    if (rc) return rc;
    int64_t *d_e = (int64_t *)h->d_stage;
    double *d_v = (double *)((char *)h->d_stage + (size_t)(cap < 1 ? 1 : cap) * 8);
    int64_t w = 0;
    rc = tri_separate(h, cap, d_e, d_v, n_violated, &w);
    if (rc) return rc;
    if (w > 0) {
        HIP_TRY(h, hipMemcpyAsync(entry_out, d_e, w * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(viol_out, d_v, w * 8, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, sdpcut_sync(h));
    *n_written = w;
    return SDPCUT_OK;
}

int sdpcut_last_timing(sdpcut_handle h, double *ms, int n)
{
    if (!h || !ms || n < 1) return SDPCUT_EINVAL;
    if (!h->timing) return sdpcut_fail(h, SDPCUT_ESTATE, "enable SDPCUT_OPT_TIMING first");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, sdpcut_sync(h));
    float a = 0.f, b = 0.f;
    if (!h->timed_score || hipEventElapsedTime(&a, h->ev[0], h->ev[1]) != hipSuccess) a = -1.f;
    if (h->timing < 2 || hipEventElapsedTime(&b, h->ev[2], h->ev[3]) != hipSuccess) b = -1.f;
    (void)hipGetLastError();   // an unrecorded pair is not an error of this library: clear the sticky code
    ms[0] = a;
    if (n > 1) ms[1] = b;
    return SDPCUT_OK;
}

int sdpcut_mfma_probe(sdpcut_handle h, const double *A, const double *B, double *C)
{
    if (!h || !A || !B || !C) return SDPCUT_EINVAL;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_stage(h, 8 * (64 + 64 + 256));
    if (rc) return rc;
    double *dA = (double *)h->d_stage, *dB = dA + 64, *dC = dB + 64;
    HIP_TRY(h, hipMemcpyAsync(dA, A, 64 * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dB, B, 64 * 8, hipMemcpyHostToDevice, h->stream));
    rc = launch_mfma_probe(h, dA, dB, dC);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(C, dC, 256 * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

} // extern "C"

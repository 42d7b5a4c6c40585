// The two candidate lists of a QCQP instance, built ON THE DEVICE (cut_select_qcqp.py:314-334, __get_vertex_cover):
//
//     objective cover  = cover(sparsity pattern of the objective)                         :320-321
//     all cover        = cover(pattern of objective + every constraint)                   :324-326
//     self._agg_list   = [s for s in all cover if s in objective cover]                   :331   (list membership: O(N^2))
//     agg_list_cons    = [s for s in all cover if s not in objective cover]               :332-333
//
// Both covers come from the device enumeration (covergen.hip), whose order is the lexicographic order of the index
// sets (no emitted set is a prefix of another): membership in the objective cover is ONE binary search per set of the
// all-cover, and the two output lists keep the all-cover's order through prefix sums.  q_50_* with 5-variable
// sub-problems (BASELINE configs[4]: 2.1e6 sets in the all-cover, a test the reference's list membership cannot finish)
// never exists on the host.
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "common.h"

namespace {

// lexicographic order of two index sets padded with -1 (a shorter set never is a prefix of a longer one in a cover)
__device__ __forceinline__ int cmp5(const int32_t *a, const int32_t *b)
{
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}

// per set of the all-cover: does it belong to the objective cover?  + the packed counters the prefix sums run over:
//   v[0] = (in ? 1 : 1 << 32)                         position in its list
//   v[1] = in && k == 2 ? 1 : in && k == 3 ? 1 << 32   position in its size class, lists "in" ...
//   v[2] = in && k == 4 ? 1 : in && k == 5 ? 1 << 32
//   v[3], v[4]  the same for "out"
__global__ void split_flag_kernel(int64_t n_all, const int32_t *set_all, const int32_t *k_all, int64_t n_obj, const int32_t *set_obj,
                                  uint8_t *flag, uint64_t *v, int64_t stride)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_all) return;
    int32_t s[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) s[a] = set_all[i * 5 + a];
    int64_t lo = 0, hi = n_obj;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cmp5(set_obj + mid * 5, s) < 0) lo = mid + 1; else hi = mid;
    }
    const bool in = lo < n_obj && cmp5(set_obj + lo * 5, s) == 0;
    flag[i] = in;
    const int k = k_all[i];
    const uint64_t one = 1ull, hiw = 1ull << 32;
    v[i] = in ? one : hiw;
    v[stride + i] = (in && k == 2) ? one : (in && k == 3) ? hiw : 0ull;
    v[2 * stride + i] = (in && k == 4) ? one : (in && k == 5) ? hiw : 0ull;
    v[3 * stride + i] = (!in && k == 2) ? one : (!in && k == 3) ? hiw : 0ull;
    v[4 * stride + i] = (!in && k == 4) ? one : (!in && k == 5) ? hiw : 0ull;
}

struct SplitSink {
    int32_t *set5[2];       // [0] in, [1] out: caller order [n][5]
    int32_t *ks[2];
    int32_t *soa[2][SDPCUT_MAX_K + 1];
    int32_t *orig[2][SDPCUT_MAX_K + 1];
    int64_t cls_n[2][SDPCUT_MAX_K + 1];
};

__global__ void split_write_kernel(int64_t n_all, const int32_t *set_all, const int32_t *k_all, const uint8_t *flag, const uint64_t *p,
                                   int64_t stride, SplitSink sk)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_all) return;
    const int part = flag[i] ? 0 : 1;
    const int k = k_all[i];
    const uint64_t p0 = p[i];
    const int64_t pos = part == 0 ? (int64_t)(p0 & 0xffffffffull) : (int64_t)(p0 >> 32);
    const uint64_t pc = p[(int64_t)(1 + 2 * part + (k >= 4 ? 1 : 0)) * stride + i];
    const int64_t pos_k = (k == 2 || k == 4) ? (int64_t)(pc & 0xffffffffull) : (int64_t)(pc >> 32);
    sk.ks[part][pos] = k;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int32_t vtx = set_all[i * 5 + a];
        sk.set5[part][pos * 5 + a] = vtx;
        if (a < k) sk.soa[part][k][(int64_t)a * sk.cls_n[part][k] + pos_k] = vtx;
    }
    sk.orig[part][k][pos_k] = (int32_t)pos;
}

} // namespace

extern "C" int sdpcut_set_candidates_cover_split(sdpcut_handle h_in, sdpcut_handle h_out, const uint8_t *adjacency_obj,
                                                 const uint8_t *adjacency_all, int32_t dim, int64_t *n_in, int64_t *n_out)
{
    if (!h_in || !h_out) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h_in);
    SDPCUT_NO_PENDING(h_out);
    sdpcut_ctx *h = h_out;
    if (h_in == h_out || h_in->device != h_out->device) return sdpcut_fail(h, SDPCUT_EINVAL, "cover_split: two handles on one device");
    if (h_in->nb_vars == 0 || h_in->nb_vars != h_out->nb_vars) return sdpcut_fail(h, SDPCUT_ESTATE, "cover_split: set_instance (same instance) on both handles first");
    if (!adjacency_obj || !adjacency_all || !n_in || !n_out) return sdpcut_fail(h, SDPCUT_EINVAL, "bad cover_split arguments");
    HIP_TRY(h, hipSetDevice(h->device));
    // 1. objective cover (into h_in, kept aside), all-cover (into h_out)
    int64_t N_o = 0, N_a = 0;
    int rc = sdpcut_set_candidates_cover(h_in, adjacency_obj, dim, 0, &N_o);
    if (rc) return sdpcut_fail(h, rc, std::string("cover_split, objective cover: ") + sdpcut_last_error(h_in));
    rc = sdpcut_set_candidates_cover(h_out, adjacency_all, dim, 0, &N_a);
    if (rc) return rc;
    HIP_TRY(h, sdpcut_sync(h_in));
    int32_t *d_obj = nullptr, *d_all = h->d_set_orig, *d_kall = h->d_k;
    uint8_t *d_flag = nullptr;
    uint64_t *d_v = nullptr, *d_p = nullptr;
    void *d_tmp = nullptr;
    // the all-cover's arrays are the source of the split: take them out of the handle before its list is replaced
    h->d_set_orig = nullptr;
    h->d_k = nullptr;
    auto cleanup = [&]() { hipFree(d_obj); hipFree(d_all); hipFree(d_kall); hipFree(d_flag); hipFree(d_v); hipFree(d_p); hipFree(d_tmp); };
    // an error exit leaves BOTH handles without a list (N = 0, nothing scored): h_out's arrays have been taken out of it above
    // and h_in still holds the whole objective cover -- a later round on either would run on half-built state
    auto fail_cleanup = [&]() { cleanup(); free_candidates(h_in); free_candidates(h_out); };
#define SPLIT_TRY(expr)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            fail_cleanup();                                                                     \
            return sdpcut_fail(h, SDPCUT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                       \
    } while (0)
    const size_t na = (size_t)(N_a < 1 ? 1 : N_a), no = (size_t)(N_o < 1 ? 1 : N_o);
    SPLIT_TRY(hipMalloc((void **)&d_obj, no * 5 * sizeof(int32_t)));
    if (N_o > 0) SPLIT_TRY(hipMemcpyAsync(d_obj, h_in->d_set_orig, (size_t)N_o * 5 * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    SPLIT_TRY(hipMalloc((void **)&d_flag, na));
    SPLIT_TRY(hipMalloc((void **)&d_v, na * 5 * sizeof(uint64_t)));
    SPLIT_TRY(hipMalloc((void **)&d_p, na * 5 * sizeof(uint64_t)));
    int64_t cnt[2][SDPCUT_MAX_K + 1] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    int64_t tot[2] = {0, 0};
    if (N_a > 0) {
        const unsigned grid = (unsigned)((N_a + 255) / 256);
        hipLaunchKernelGGL(split_flag_kernel, dim3(grid), dim3(256), 0, h->stream, N_a, d_all, d_kall, N_o, d_obj, d_flag, d_v, N_a);
        SPLIT_TRY(hipGetLastError());
        size_t tb = 0;
        SPLIT_TRY(rocprim::exclusive_scan(nullptr, tb, d_v, d_p, 0ull, (size_t)N_a, rocprim::plus<uint64_t>(), h->stream));
        SPLIT_TRY(hipMalloc(&d_tmp, tb < 256 ? 256 : tb));
        for (int c = 0; c < 5; ++c)
            SPLIT_TRY(rocprim::exclusive_scan(d_tmp, tb, d_v + (size_t)c * N_a, d_p + (size_t)c * N_a, 0ull, (size_t)N_a,
                                              rocprim::plus<uint64_t>(), h->stream));
        // totals = prefix of the last element + its own contribution
        uint64_t lastp[5], lastv[5];
        for (int c = 0; c < 5; ++c) {
            SPLIT_TRY(hipMemcpyAsync(&lastp[c], d_p + (size_t)c * N_a + (N_a - 1), 8, hipMemcpyDeviceToHost, h->stream));
            SPLIT_TRY(hipMemcpyAsync(&lastv[c], d_v + (size_t)c * N_a + (N_a - 1), 8, hipMemcpyDeviceToHost, h->stream));
        }
        SPLIT_TRY(sdpcut_sync(h));
        uint64_t t5[5];
        for (int c = 0; c < 5; ++c) t5[c] = lastp[c] + lastv[c];
        tot[0] = (int64_t)(t5[0] & 0xffffffffull); tot[1] = (int64_t)(t5[0] >> 32);
        for (int part = 0; part < 2; ++part) {
            cnt[part][2] = (int64_t)(t5[1 + 2 * part] & 0xffffffffull); cnt[part][3] = (int64_t)(t5[1 + 2 * part] >> 32);
            cnt[part][4] = (int64_t)(t5[2 + 2 * part] & 0xffffffffull); cnt[part][5] = (int64_t)(t5[2 + 2 * part] >> 32);
        }
    }
    // 2. the two lists
    SPLIT_TRY(sdpcut_sync(h));      // (the objective cover has left h_in's arrays)
    rc = alloc_candidates(h_in, tot[0], cnt[0], 0);
    if (rc) { fail_cleanup(); return sdpcut_fail(h, rc, std::string("cover_split: ") + sdpcut_last_error(h_in)); }
    rc = alloc_candidates(h_out, tot[1], cnt[1], 0);
    if (rc) { fail_cleanup(); return rc; }
    if (N_a > 0) {
        SplitSink sk{};
        sdpcut_ctx *hh[2] = {h_in, h_out};
        for (int part = 0; part < 2; ++part) {
            sk.set5[part] = hh[part]->d_set_orig;
            sk.ks[part] = hh[part]->d_k;
            for (int s = 2; s <= SDPCUT_MAX_K; ++s) {
                sk.soa[part][s] = hh[part]->bucket[s].d_set;
                sk.orig[part][s] = hh[part]->bucket[s].d_orig;
                sk.cls_n[part][s] = cnt[part][s];
            }
        }
        hipLaunchKernelGGL(split_write_kernel, dim3((unsigned)((N_a + 255) / 256)), dim3(256), 0, h->stream, N_a, d_all, d_kall, d_flag, d_p,
                           N_a, sk);
        SPLIT_TRY(hipGetLastError());
        SPLIT_TRY(sdpcut_sync(h));
    }
    cleanup();
#undef SPLIT_TRY
    *n_in = tot[0];
    *n_out = tot[1];
    return SDPCUT_OK;
}

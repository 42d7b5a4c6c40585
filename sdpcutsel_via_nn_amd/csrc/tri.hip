// Triangle-inequality separation (SURVEY.md section 8 f, row 3; reference
// cut_select_qp.py:799-863).  Per round every retained triple i1<i2<i3 yields four
// violations; the violated ones (>= 1e-7) are ranked by (density desc, violation desc), ties
// in entry order 4*triple + type (Python's stable sort), and the head is returned.  The caller
// consumes at most 10 000 of the ~1e6 entries (_TRI_CUTS_PER_ROUND_MAX): the head comes from the
// radix select of topk.hip over the composite keys, not from a sort of all of them.
#include <cstring>

#include "common.h"

#define TRI_VIOL_THRES 1e-7     /* _THRES_TRI_VIOL, cut_select_qp.py:33 */

// key: bit 63 = density 3, low bits = the positive violation's IEEE image (positive doubles
// order like integers); 0 = not violated.  Violations are <= 2, so bit 63 is free.
__global__ __launch_bounds__(256) void tri_viol_kernel(int64_t T, const int32_t *tri, const uint8_t *dense3,
                                                       const double *vars, int32_t nv, int64_t L, uint64_t *key,
                                                       uint32_t *val, int64_t *counters)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int nviol = 0;
    if (t < T) {
#pragma clang fp contract(off)
        const int32_t a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
        const int32_t ra = nv * a - (a * (a + 1)) / 2, rb = nv * b - (b * (b + 1)) / 2;
        const double X1 = vars[ra + b], X2 = vars[ra + c], X4 = vars[rb + c];      // X_slice[1], [2], [4]
        const double x0 = vars[L + a], x1 = vars[L + b], x2 = vars[L + c];
        double v[4];
        v[0] = X1 + X2 - X4 - x0;                                                  // :836-839, left to right
        v[1] = X1 - X2 + X4 - x1;
        v[2] = -X1 + X2 + X4 - x2;
        v[3] = -X1 - X2 - X4 + (((0.0 + x0) + x1) + x2) - 1.0;
        const uint64_t hi = dense3[t] ? 0x8000000000000000ull : 0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool viol = v[k] >= TRI_VIOL_THRES;                              // :841
            key[4 * t + k] = viol ? (hi | (uint64_t)__double_as_longlong(v[k])) : 0ull;
            val[4 * t + k] = (uint32_t)(4 * t + k);
            nviol += viol;
        }
    }
    for (int off = 32; off > 0; off >>= 1) nviol += __shfl_xor(nviol, off);
    if ((threadIdx.x & 63) == 0 && nviol) atomicAdd((unsigned long long *)&counters[0], (unsigned long long)nviol);
}

int tri_preprocess(sdpcut_ctx *h, const uint8_t *adjacency, int64_t *n_triples)
{
    const int n = h->nb_vars;
    std::vector<int32_t> tri;
    std::vector<uint8_t> d3;
    auto adj = [&](int i, int j) { return adjacency[(size_t)i * n + j] != 0; };
    for (int i1 = 0; i1 < n; ++i1)
        for (int i2 = i1 + 1; i2 < n; ++i2)
            for (int i3 = i2 + 1; i3 < n; ++i3) {
                const int dens = (int)adj(i1, i2) + (int)adj(i1, i3) + (int)adj(i2, i3);   // :812
                if (dens >= 2) {                                                          // _THRES_TRI_DENSE
                    tri.push_back(i1); tri.push_back(i2); tri.push_back(i3);
                    d3.push_back(dens == 3);
                }
            }
    const int64_t T = (int64_t)d3.size();
    if (4 * T > 0x7fffffffLL) return sdpcut_fail(h, SDPCUT_EINVAL, "too many triangle inequalities");
    (void)hipFree(h->d_tri); (void)hipFree(h->d_tri_dense3);
    h->d_tri = nullptr; h->d_tri_dense3 = nullptr; h->n_tri = 0;
    if (T > 0) {
        HIP_TRY(h, hipMalloc((void **)&h->d_tri, (size_t)T * 3 * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc((void **)&h->d_tri_dense3, (size_t)T));
        HIP_TRY(h, hipMemcpy(h->d_tri, tri.data(), (size_t)T * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_tri_dense3, d3.data(), (size_t)T, hipMemcpyHostToDevice));
    }
    h->n_tri = T;
    h->tri_host = tri;
    h->tri_dense_host = d3;
    if (n_triples) *n_triples = T;
    return 0;
}

int tri_separate(sdpcut_ctx *h, int64_t max_out, int64_t *d_entry_out, double *d_viol_out, int64_t *n_violated,
                 int64_t *n_written)
{
    const int64_t T = h->n_tri, E = 4 * T;
    int64_t cnt[5] = {0, 0, 0, 0, 0};
    if (max_out > 16384) return sdpcut_fail(h, SDPCUT_EINVAL, "tri_separate: at most 16384 entries (the reference takes <= 10000)");
    if (E > 0) {
        int rc = ensure_rank_ws(h, E > h->N ? E : h->N);      // (the selection below sizes the key array for the candidate list as well: it must not move)
        if (rc) return rc;
        h->last_total = -1;
        HIP_TRY(h, hipMemsetAsync(h->d_counters, 0, 8 * sizeof(int64_t), h->stream));
        hipLaunchKernelGGL(tri_viol_kernel, dim3((int)((T + 255) / 256)), dim3(256), 0, h->stream, T, h->d_tri,
                           h->d_tri_dense3, h->d_vars, h->nb_vars, h->L, h->d_key_a, h->d_val_a, h->d_counters);
        HIP_TRY(h, hipGetLastError());
        if (max_out > 0) {
            rc = topk_select_keys_on_device(h, E, max_out, d_entry_out, d_viol_out, cnt);
            if (rc) return rc;
            if (cnt[4]) {
                // a bounded wait of the fused selection expired (the GPU shared with a kernel that kept its workgroups
                // from starting): the same selection with one launch per digit has no waits and always answers
                const bool fused = h->fused_tail;
                h->fused_tail = false;
                rc = topk_select_keys_on_device(h, E, max_out, d_entry_out, d_viol_out, cnt);
                h->fused_tail = fused;
                ++h->stat_fallbacks;
                if (rc) return rc;
                if (cnt[4]) return sdpcut_fail(h, SDPCUT_EHIP, "tri_separate: selection void");
            }
        } else {
            HIP_TRY(h, hipMemcpyAsync(cnt, h->d_counters, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, sdpcut_sync(h));
        }
    }
    const int64_t w = cnt[0] < max_out ? cnt[0] : max_out;
    if (n_violated) *n_violated = cnt[0];
    if (n_written) *n_written = w;
    return 0;
}

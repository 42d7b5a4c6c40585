// sdpcut_set_candidates_philox: the candidate list of the C4 workload generated in device memory
// (no host array of N index sets exists at any point), see philox.h.
#include "common.h"
#include "philox.h"

__global__ __launch_bounds__(256) void philox_sets_kernel(uint64_t seed, uint64_t first_id, int64_t n_cand, int nv,
                                                          int k, int32_t *soa, int32_t *orig, int32_t *set5,
                                                          int32_t *ks)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_cand) return;
    int32_t s[5];
    philox_index_set(seed, first_id + (uint64_t)i, nv, k, s);
    for (int a = 0; a < k; ++a) soa[(int64_t)a * n_cand + i] = s[a];       // lane-contiguous per index
    orig[i] = (int32_t)i;
#pragma unroll
    for (int a = 0; a < 5; ++a) set5[i * 5 + a] = s[a];
    ks[i] = k;
}

extern "C" int sdpcut_set_candidates_philox(sdpcut_handle h, int32_t k, int64_t N, uint64_t seed, int64_t first_id)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (h->nb_vars == 0) return sdpcut_fail(h, SDPCUT_ESTATE, "set_instance first");
    if (k < 2 || k > SDPCUT_MAX_K) return sdpcut_fail(h, SDPCUT_EINVAL, "k must be 2..5");
    if (N < 0 || N > 0x7fffffffLL || first_id < 0) return sdpcut_fail(h, SDPCUT_EINVAL, "bad candidate count / first id");
    if (h->nb_vars < 2 * k) return sdpcut_fail(h, SDPCUT_EINVAL, "the generator needs nb_vars >= 2 k");
    int64_t cnt[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
    cnt[k] = N;
    int rc = alloc_candidates(h, N, cnt, first_id);      // global ids = generator ids
    if (rc) return rc;
    if (N == 0) return SDPCUT_OK;
    const Bucket &b = h->bucket[k];
    hipLaunchKernelGGL(philox_sets_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, seed,
                       (uint64_t)first_id, N, (int)h->nb_vars, (int)k, b.d_set, b.d_orig, h->d_set_orig, h->d_k);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

// Index sets of `count` candidates by LOCAL index, device -> host (row i = 5 entries padded with -1):
// what a host needs to name the few thousand selected candidates of a list that only exists on the
// device (generated or enumerated there).
__global__ void gather_sets_kernel(int64_t count, const int64_t *idx, int64_t n, const int32_t *set5, const int32_t *ks,
                                   int32_t *out5, int32_t *out_k)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t c = idx[i];
    const bool ok = c >= 0 && c < n;
    for (int a = 0; a < 5; ++a) out5[i * 5 + a] = ok ? set5[c * 5 + a] : -1;
    out_k[i] = ok ? ks[c] : 0;
}

extern "C" int sdpcut_get_candidates(sdpcut_handle h, int64_t count, const int64_t *idx, int32_t *set_inds_out,
                                     int32_t *ks_out)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (count < 0 || (count > 0 && (!idx || !set_inds_out || !ks_out)))
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad get_candidates arguments");
    if (!h->d_set_orig) return sdpcut_fail(h, SDPCUT_ESTATE, "no candidate list");
    if (count == 0) return SDPCUT_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t c = (size_t)count;
    int rc = ensure_stage(h, c * (8 + 20 + 4));
    if (rc) return rc;
    int64_t *d_idx = (int64_t *)h->d_stage;
    int32_t *d_out = (int32_t *)(d_idx + c), *d_k = d_out + c * 5;
    HIP_TRY(h, hipMemcpyAsync(d_idx, idx, c * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(gather_sets_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, count, d_idx,
                       h->N, h->d_set_orig, h->d_k, d_out, d_k);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(set_inds_out, d_out, c * 20, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(ks_out, d_k, c * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return SDPCUT_OK;
}

// Semidefinite vertex cover enumerated ON THE DEVICE, straight into the handle's candidate arrays
// (SURVEY.md section 8 f row 1: "emitting set_inds directly in device memory"; lifts the 4e6 cap
// of cut_select_qp.py:117-120 -- spar125-075-1 with dim 5 has 12.8e6 candidates of sizes 2..5).
//
// Same cover and same ORDER as csrc/cover.cpp / the reference's nested loops
// (cut_select_qp.py:399-524, ch_ext = 0): every clique of size dim, plus every clique of size
// 2..dim-1 that no vertex -- larger or smaller -- extends.  No emitted set is a prefix of another,
// so the reference's DFS order is the lexicographic order of the sets, and the output position of
// a set is (sets emitted under lexicographically smaller edges) + (sets under smaller third
// vertices of its edge) + (its rank inside its triangle's subtree).  That decomposes:
//
//   count pass   one wave per edge (i1,i2) of the sparsity graph, one lane per forward triangle
//                (i1,i2,i3): the lane walks the triangle's subtree (<= 2 more levels, bitset
//                intersections in registers) and counts its sets per size; wave reduction
//                -> per-edge totals;
//   scan         exclusive prefix sums over the edges (one workgroup) -> first position of every
//                edge, in the whole list and in each size class;
//   write pass   same walk; lanes get their offsets from a wave prefix sum and store the sets in
//                list order (d_set_orig / d_k) AND in the per-size SoA buckets the score kernels
//                read (d_set / d_orig) -- the host never holds the list.
#include <vector>

#include "common.h"

namespace {

template <int W>
struct Bits {
    uint64_t w[W];
};

template <int W>
__device__ __forceinline__ Bits<W> load_row(const uint64_t *adj, int v)
{
    Bits<W> b;
#pragma unroll
    for (int u = 0; u < W; ++u) b.w[u] = adj[(size_t)v * W + u];
    return b;
}
template <int W>
__device__ __forceinline__ Bits<W> band(const Bits<W> &a, const Bits<W> &b)
{
    Bits<W> c;
#pragma unroll
    for (int u = 0; u < W; ++u) c.w[u] = a.w[u] & b.w[u];
    return c;
}
// bits strictly above v / strictly below v
template <int W>
__device__ __forceinline__ Bits<W> above(const Bits<W> &a, int v)
{
    Bits<W> c;
    const int t = v + 1;
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const uint64_t m = (u < (t >> 6)) ? 0ull : (u == (t >> 6) ? (~0ull << (t & 63)) : ~0ull);
        c.w[u] = a.w[u] & m;
    }
    return c;
}
template <int W>
__device__ __forceinline__ bool any_below(const Bits<W> &a, int v)
{
    uint64_t acc = 0;
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const uint64_t m = (u < (v >> 6)) ? ~0ull : (u == (v >> 6) ? ((v & 63) ? (~0ull >> (64 - (v & 63))) : 0ull) : 0ull);
        acc |= a.w[u] & m;
    }
    return acc != 0;
}
template <int W>
__device__ __forceinline__ int popcount(const Bits<W> &a)
{
    int c = 0;
#pragma unroll
    for (int u = 0; u < W; ++u) c += __popcll(a.w[u]);
    return c;
}
// index of the m-th (0-based) set bit; m < popcount
template <int W>
__device__ __forceinline__ int nth_bit(const Bits<W> &a, int m)
{
    int v = -1;
    bool done = false;
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const int c = __popcll(a.w[u]);
        if (!done && m < c) {
            uint64_t x = a.w[u];
            for (int t = 0; t < m; ++t) x &= x - 1;
            v = (u << 6) + __builtin_ctzll(x);
            done = true;
        }
        m -= c;
    }
    return v;
}

struct Sink {          // write-pass destinations
    int32_t *set5, *ks;                       // list order
    int32_t *soa[SDPCUT_MAX_K + 1], *orig[SDPCUT_MAX_K + 1];
    int64_t cls_n[SDPCUT_MAX_K + 1];          // size of every size class (SoA stride)
};

// Walks the subtree of triangle (i1,i2,i3).  WRITE = false: counts into cnt[size]; WRITE = true:
// stores every set at list position g (advancing) and class positions p[size] (advancing).
template <int W, bool WRITE>
__device__ __forceinline__ void walk_triangle(const uint64_t *adj, int dim, int i1, int i2, int i3, const Bits<W> &c3,
                                              int (&cnt)[SDPCUT_MAX_K + 1], const Sink *sk, int64_t &g,
                                              int64_t (&p)[SDPCUT_MAX_K + 1])
{
    auto emit = [&](int size, int a4, int a5) {
        if constexpr (WRITE) {
            const int32_t s[5] = {i1, i2, i3, size > 3 ? a4 : -1, size > 4 ? a5 : -1};
#pragma unroll
            for (int a = 0; a < 5; ++a) sk->set5[g * 5 + a] = s[a];
            sk->ks[g] = size;
            for (int a = 0; a < size; ++a) sk->soa[size][(int64_t)a * sk->cls_n[size] + p[size]] = s[a];
            sk->orig[size][p[size]] = (int32_t)g;
            ++g;
            ++p[size];
        } else {
            ++cnt[size];
        }
    };
    if (dim == 3) { emit(3, -1, -1); return; }
    Bits<W> f4 = above<W>(c3, i3);
    if (popcount<W>(f4) == 0) {
        if (!any_below<W>(c3, i3)) emit(3, -1, -1);
        return;
    }
#pragma unroll 1
    for (int u4 = 0; u4 < W; ++u4) {
        uint64_t bits4 = f4.w[u4];
        while (bits4) {
            const int i4 = (u4 << 6) + __builtin_ctzll(bits4);
            bits4 &= bits4 - 1;
            if (dim == 4) { emit(4, i4, -1); continue; }
            const Bits<W> c4 = band<W>(c3, load_row<W>(adj, i4));
            const Bits<W> f5 = above<W>(c4, i4);
            if (popcount<W>(f5) == 0) {
                if (!any_below<W>(c4, i4)) emit(4, i4, -1);
                continue;
            }
#pragma unroll 1
            for (int u5 = 0; u5 < W; ++u5) {
                uint64_t bits5 = f5.w[u5];
                while (bits5) {
                    const int i5 = (u5 << 6) + __builtin_ctzll(bits5);
                    bits5 &= bits5 - 1;
                    emit(5, i4, i5);
                }
            }
        }
    }
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total)
{
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    total = __shfl(inc, 63);
    return inc - v;
}

// per_edge: [E][5] = {total, n2, n3, n4, n5}; WRITE reads off: [E][5] exclusive prefix sums
template <int W, bool WRITE>
__global__ __launch_bounds__(256) void cover_kernel(const uint64_t *adj, const int32_t *edges, int64_t n_edges, int dim,
                                                    int32_t *per_edge, const int64_t *off, Sink sk)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= n_edges) return;          // wave-uniform
    const int i1 = edges[2 * e], i2 = edges[2 * e + 1];
    const Bits<W> c2 = band<W>(load_row<W>(adj, i1), load_row<W>(adj, i2));
    const Bits<W> f3 = above<W>(c2, i2);
    const int n3 = popcount<W>(f3);
    int64_t base = WRITE ? off[5 * e] : 0;
    int64_t cbase[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
    if constexpr (WRITE) {
        for (int s = 2; s <= SDPCUT_MAX_K; ++s) cbase[s] = off[5 * e + s - 1];
    }
    int tot[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
    if (n3 == 0) {
        // no larger vertex extends the edge: a 2-set iff no smaller one does either
        if (!any_below<W>(c2, i2)) {
            tot[2] = 1;
            if (WRITE && lane == 0) {
                const int64_t g = base, p = cbase[2];
                sk.set5[g * 5 + 0] = i1; sk.set5[g * 5 + 1] = i2;
                sk.set5[g * 5 + 2] = -1; sk.set5[g * 5 + 3] = -1; sk.set5[g * 5 + 4] = -1;
                sk.ks[g] = 2;
                sk.soa[2][p] = i1;
                sk.soa[2][sk.cls_n[2] + p] = i2;
                sk.orig[2][p] = (int32_t)g;
            }
        }
    }
    for (int r0 = 0; r0 < n3; r0 += 64) {
        const int m = r0 + lane;
        const bool live = m < n3;
        int cnt[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
        int i3 = 0;
        Bits<W> c3 = c2;
        int64_t g = 0, p[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
        if (live) {
            i3 = nth_bit<W>(f3, m);
            c3 = band<W>(c2, load_row<W>(adj, i3));
            walk_triangle<W, false>(adj, dim, i1, i2, i3, c3, cnt, nullptr, g, p);
        }
        const int mine = cnt[2] + cnt[3] + cnt[4] + cnt[5];
        int round_total;
        const int before = wave_excl_scan(mine, lane, round_total);
        int cls_before[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0}, cls_total[SDPCUT_MAX_K + 1] = {0, 0, 0, 0, 0, 0};
        for (int s = 3; s <= dim; ++s) cls_before[s] = wave_excl_scan(cnt[s], lane, cls_total[s]);
        if constexpr (WRITE) {
            if (live) {
                g = base + before;
                for (int s = 3; s <= dim; ++s) p[s] = cbase[s] + cls_before[s];
                walk_triangle<W, true>(adj, dim, i1, i2, i3, c3, cnt, &sk, g, p);
            }
        }
        base += round_total;
        for (int s = 3; s <= dim; ++s) { cbase[s] += cls_total[s]; tot[s] += cls_total[s]; }
    }
    if (!WRITE && lane == 0) {
        per_edge[5 * e] = tot[2] + tot[3] + tot[4] + tot[5];
        for (int s = 2; s <= SDPCUT_MAX_K; ++s) per_edge[5 * e + s - 1] = tot[s];
    }
}

// exclusive prefix sums of the five per-edge columns; off[5 E .. 5 E + 4] = the grand totals
__global__ __launch_bounds__(1024) void cover_scan_kernel(const int32_t *per_edge, int64_t n_edges, int64_t *off)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t chunk = (n_edges + 1023) / 1024;
    const int64_t lo = t * chunk < n_edges ? t * chunk : n_edges, hi = (lo + chunk < n_edges) ? lo + chunk : n_edges;
    for (int col = 0; col < 5; ++col) {
        int64_t s = 0;
        for (int64_t e = lo; e < hi; ++e) s += per_edge[5 * e + col];
        part[t] = s;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {          // inclusive Hillis-Steele scan of the partial sums
            const int64_t v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        int64_t run = part[t] - s;
        for (int64_t e = lo; e < hi; ++e) {
            off[5 * e + col] = run;
            run += per_edge[5 * e + col];
        }
        if (t == 1023) off[5 * n_edges + col] = part[1023];
        __syncthreads();
    }
}

template <int W>
int run_cover(sdpcut_ctx *h, const std::vector<uint64_t> &adj_host, const std::vector<int32_t> &edges, int dim,
              int64_t max_subs, int64_t *count_out)
{
    const int64_t E = (int64_t)(edges.size() / 2);
    uint64_t *d_adj = nullptr;
    int32_t *d_edges = nullptr, *d_per = nullptr;
    int64_t *d_off = nullptr;
    auto cleanup = [&]() { hipFree(d_adj); hipFree(d_edges); hipFree(d_per); hipFree(d_off); };
#define COVER_TRY(expr)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            cleanup();                                                                          \
            return sdpcut_fail(h, SDPCUT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                       \
    } while (0)
    COVER_TRY(hipMalloc((void **)&d_adj, adj_host.size() * 8));
    COVER_TRY(hipMalloc((void **)&d_edges, (size_t)(E < 1 ? 1 : E) * 8));
    COVER_TRY(hipMalloc((void **)&d_per, (size_t)(E < 1 ? 1 : E) * 20));
    COVER_TRY(hipMalloc((void **)&d_off, (size_t)(E + 1) * 40));
    COVER_TRY(hipMemcpy(d_adj, adj_host.data(), adj_host.size() * 8, hipMemcpyHostToDevice));
    if (E > 0) COVER_TRY(hipMemcpy(d_edges, edges.data(), (size_t)E * 8, hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((E + 3) / 4);
    Sink sk{};
    if (E > 0) hipLaunchKernelGGL((cover_kernel<W, false>), dim3(grid), dim3(256), 0, h->stream, d_adj, d_edges, E, dim, d_per,
                                  (const int64_t *)nullptr, sk);
    hipLaunchKernelGGL(cover_scan_kernel, dim3(1), dim3(1024), 0, h->stream, d_per, E, d_off);
    COVER_TRY(hipGetLastError());
    int64_t totals[5] = {0, 0, 0, 0, 0};
    COVER_TRY(hipMemcpyAsync(totals, d_off + 5 * E, 40, hipMemcpyDeviceToHost, h->stream));
    COVER_TRY(sdpcut_sync(h));
    *count_out = totals[0];
    if (totals[0] > 0x7fffffffLL) { cleanup(); return sdpcut_fail(h, SDPCUT_EINVAL, "cover has more than 2^31 - 1 candidates"); }
    if (max_subs > 0 && totals[0] >= max_subs) { cleanup(); return SDPCUT_OK; }     // count only (the reference's guard)
    int64_t cnt[SDPCUT_MAX_K + 1] = {0, 0, totals[1], totals[2], totals[3], totals[4]};
    int rc = alloc_candidates(h, totals[0], cnt, 0);
    if (rc) { cleanup(); return rc; }
    if (totals[0] > 0) {
        sk.set5 = h->d_set_orig;
        sk.ks = h->d_k;
        for (int s = 2; s <= SDPCUT_MAX_K; ++s) {
            sk.soa[s] = h->bucket[s].d_set;
            sk.orig[s] = h->bucket[s].d_orig;
            sk.cls_n[s] = cnt[s];
        }
        hipLaunchKernelGGL((cover_kernel<W, true>), dim3(grid), dim3(256), 0, h->stream, d_adj, d_edges, E, dim, d_per,
                           (const int64_t *)d_off, sk);
        COVER_TRY(hipGetLastError());
        COVER_TRY(sdpcut_sync(h));
    }
    cleanup();
#undef COVER_TRY
    return SDPCUT_OK;
}

} // namespace

extern "C" int sdpcut_set_candidates_cover(sdpcut_handle h, const uint8_t *adjacency, int32_t dim, int64_t max_subs,
                                           int64_t *count_out)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (h->nb_vars == 0) return sdpcut_fail(h, SDPCUT_ESTATE, "set_instance first");
    if (!adjacency || dim < 3 || dim > SDPCUT_MAX_K || !count_out || max_subs < 0)
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad cover arguments (dim must be 3..5)");
    const int n = h->nb_vars;
    if (n > 1024) return sdpcut_fail(h, SDPCUT_EINVAL, "device cover enumeration supports nb_vars <= 1024");
    HIP_TRY(h, hipSetDevice(h->device));
    const int W = n <= 64 ? 1 : n <= 128 ? 2 : n <= 256 ? 4 : 16;
    std::vector<uint64_t> adj((size_t)n * W, 0);
    std::vector<int32_t> edges;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (i != j && (adjacency[(size_t)i * n + j] || adjacency[(size_t)j * n + i])) {
                adj[(size_t)i * W + (j >> 6)] |= 1ull << (j & 63);
                if (i < j) { edges.push_back(i); edges.push_back(j); }      // lexicographic by construction
            }
    *count_out = 0;
    switch (W) {
    case 1: return run_cover<1>(h, adj, edges, dim, max_subs, count_out);
    case 2: return run_cover<2>(h, adj, edges, dim, max_subs, count_out);
    case 4: return run_cover<4>(h, adj, edges, dim, max_subs, count_out);
    default: return run_cover<16>(h, adj, edges, dim, max_subs, count_out);
    }
}

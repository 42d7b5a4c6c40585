// Scoring kernels of libsdpcut_hip.so -- hand-written for gfx950 (CDNA4, wave64).
//
// One launch per candidate size k scores every k-variable candidate at the current LP
// point (reference: the loop bodies of _sel_eigcut_by_ordering_on_measure,
// cut_select_qp.py:570-582 and :642-648, plus NNs.so and numpy.linalg.eigvalsh under them):
//
//   phase A  lane = candidate: read the index set (SoA, coalesced), gather x_rho / X_rho /
//            Q_rho from the HBM-resident, cache-hot tables, derive max_elem / Q_slice / S,
//            Jacobi lambda_min of the lifted matrix in registers, stage the mapminmax'ed MLP
//            inputs in LDS (feature-major, one 64-candidate strip per wave);
//   phase B  the MLP on the matrix cores: H^T = tansig(W * X^T + b) with
//            v_mfma_f64_16x16x4_f64, neurons on the M axis and candidates on the N axis, so
//            that the C/D fragment of one layer IS the B fragment of the next (row =
//            (lane>>4) + 4*reg is exactly k-step reg of row tile t) -- activations never leave
//            registers; weights are pre-packed host-side into A-fragment order and streamed
//            from L2 as coalesced 512-B wave loads; tansig runs on the VALU between MFMAs.
//
// A second, deliberately simple kernel (lane = candidate, reference operation order, no
// MFMA) exists for cross-checking and A/B timing (SDPCUT_KERNEL_SIMPLE).
#include <hip/hip_ext.h>

#include <type_traits>

#include "common.h"
#include "jacobi.h"
#include "gather.h"
#include "libm_exp.h"
#include "topk_dev.h"

typedef double d4 __attribute__((ext_vector_type(4)));

struct ScoreArgs {
    const int32_t *set;   // SoA [K][n]
    const int32_t *orig;  // [n]
    int64_t n;
    // score_mfma_kernel: candidates per strip of a wave, 64 or 32 (launch_score_k; see the kernel)
    int32_t strip;
    // Balanced tail round (r4, set_balanced_tail): candidates [0, rr_end) go round-robin in whole strips as before; the rest --
    // less than one strip per wave -- is split EVENLY: wave g takes tail_hi (g < tail_nhi) or tail_lo column tiles of 16
    // candidates from rr_end on.  rr_end = n, tail_hi = tail_lo = 0: no tail (every list but the ones the rule below picks).
    int64_t rr_end;
    int64_t tail_nhi;
    int32_t tail_hi, tail_lo;
    const double *vars;   // [L + nv]: X packed | x
    const double *Q;      // [L]
    int32_t nv;
    int64_t L;
    double *eig_out;      // [N] caller order
    double *obj_out;      // [N]
    uint32_t flags;
    // Leading-digit histogram of the top-k selection that follows (topk_dev.h; tk == nullptr: off): the
    // kernel that produces the scores also counts the members of the selection's class by the first radix
    // digit of their keys -- in LDS per workgroup, flushed with no-return atomics at its end (no ticket,
    // nobody waits) -- together with the violated / positive counters.  The selection then starts at its
    // second digit and builds the keys from the scores as it reads them: the separate key pass
    // (tk_keys_kernel, 17.5 us per round) is gone.  Candidates outside the class (key 0) are not counted:
    // the selection never looks below the class.
    TopkWs *tk;
    int tk_mode;           // TK_MODE_FEAS / OPT / STRONG: the kernel's FUSE template argument
    int32_t spread;        // the four waves of a workgroup take strips from four distant quarters of the list (see score_mfma_body)
    int32_t pf_mloc;       // fine histogram of the class (topk_dev.h): a workgroup reports its table down to its pf_mloc-th largest member; 0: off
    // optional: += number of candidates with obj_improve > 0 and lambda_min < -1e-15 (the "strong" class
    // of the combined strategy, cut_select_qp.py:607-613); lets the selection that follows pick its
    // regime on the device.  Needs both flags.
    int64_t *strong_out;
    NetDev net;
};

// tansig as MATLAB defines it (neural_net_3D.m:77-79): a = 2 / (1 + exp(-2 n)) - 1
__device__ __forceinline__ double tansig_lib(double n)
{
    return 2.0 / (exp(-2.0 * n) + 1.0) - 1.0;
}

// The same formula with a branch-free exp and reciprocal: 23 VALU instructions instead of the
// ~36 of the library route (every VALU instruction costs ~2-2.5 ns per wave on gfx950 whatever
// its type, v_rcp_f64 ~7 ns: profiles/r01_ubench_fp64_instruction_costs.txt -- the COUNT is
// what matters).
//
// exp(-2n): with y = -2n,  exp(y) = 2^k * exp(r/8)^8,  k = rint(y log2 e),  r/8 = y/8 - k ln2/8
// (|r/8| <= ln2/16 = 0.0433), three squarings.
//  * one-constant reduction: fl(ln2/8) is off by <= 7e-18, so r/8 is off by <= |k| 7e-18; where
//    tansig is sensitive to exp (|k| <= 40, sensitivity 2e/(1+e)^2 <= 1/2) that is <= 3e-16 in
//    the result, in saturation the sensitivity kills it;
//  * degree-7 near-minimax polynomial (truncated Chebyshev series of exp on |r| <= ln2/16;
//    coefficients computed in 60-digit arithmetic): max relative error 5e-18, the accuracy of the
//    degree-8 Taylor polynomial with one FMA less;
//  * only the upper clamp is needed (y8max = 88 keeps exp finite; towards -inf ldexp underflows
//    to 0 and tansig saturates at +1 by itself).
//  * k is rounded with the 1.5 * 2^52 trick: the low dword of t IS k as an int32 (|k| < 2^31, i.e.
//    |y| < 1.4e9 -- pre-activations of these networks stay below 1e3), which saves the
//    double->int conversion; with the constant lowered by SHIFT the dword holds k - SHIFT, so
//    exp(y) / 2 (SHIFT = 1, what tansig4/tansig8 want) costs nothing extra and is exact.
// fmin() goes through the IEEE quieting rule (v_max_f64 x, x in front of the v_min_f64): the operand
// here is an accumulator that cannot be a signalling NaN, so take the bare instruction (one issue
// slot per activation, 150 activations per candidate)
__device__ __forceinline__ double min_f64_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double max_f64_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// CLAMP = false: the caller guarantees y8_in <= y8max (NetDev::unclamped_ok: the network's
// pre-activations are bounded, see sdpcut_set_network) -- one instruction less per activation.
// Degree of the exp polynomial.  7 (default): tansig to 8e-16, obj_improve to <= 1e-11 of the CPU path on the
// synthetic workload.  6 saves one FMA per activation (-4.5 us of 333 on 1e6 three-variable candidates) but
// raises the score error to 2-5e-10 relative (tools/accuracy.py) -- inside BASELINE's 1e-6, too close to the
// 1e-9 this repository tests to; measured and left off.
#ifndef SDPCUT_EXP_DEGREE
#define SDPCUT_EXP_DEGREE 7
#endif
template <int SHIFT, bool CLAMP = true>
__device__ __forceinline__ double exp_y8_scaled(double y8_in, double y8max)     // exp(8 y8_in) / 2^SHIFT
{
    constexpr double MAGIC = 0x1.8p52 - (double)SHIFT;
    const double y8 = CLAMP ? min_f64_raw(y8_in, y8max) : y8_in;
    const double t = fma(y8, 11.541560327111707259, MAGIC);         // 8 log2 e
    const double k = t - MAGIC;
    const double r = fma(k, -0x1.62e42fefa39efp-4, y8);              // fl(ln2 / 8)
#if SDPCUT_EXP_DEGREE == 7
    double p = 0x1.a02041015378fp-13;
    p = fma(p, r, 0x1.6c1d00cea5bf1p-10);
    p = fma(p, r, 0x1.111111080fc42p-7);
    p = fma(p, r, 0x1.5555554653263p-5);
    p = fma(p, r, 0x1.5555555555689p-3);
    p = fma(p, r, 0x1.0000000000171p-1);
#else   // degree 6, weighted minimax of (exp(r) - 1 - r) / r^2 (tools/exp_poly.py): relative error 1.5e-15
    double p = 0x1.6c0ed7b92eac2p-10;
    p = fma(p, r, 0x1.1115b77b92f70p-7);
    p = fma(p, r, 0x1.5555558fc88efp-5);
    p = fma(p, r, 0x1.55555548f8ee9p-3);
    p = fma(p, r, 0x1.fffffffffee2fp-2);
#endif
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    p = p * p;
    p = p * p;
    p = p * p;
    return ldexp(p, __double2loint(t));
}

// exp(8 y8) / 2^SHIFT + addend for BOUNDED arguments (|8 y8| <= 80, NetDev::unclamped_ok): the power of two goes
// into the exponent field of p^4 with one integer add -- p^4 lies in [0.65, 1.47] and |k| <= 117, so the field
// neither overflows nor reaches the denormals -- and the last squaring, the scaling and the addend become ONE
// fma: 14 instead of 16 issue slots per activation, one rounding less.
#ifndef SDPCUT_EXP_SCALE_LDEXP
#define SDPCUT_EXP_SCALE_LDEXP 1
#endif
template <int SHIFT>
__device__ __forceinline__ double exp_y8_plus_bounded(double y8, double addend)
{
    constexpr double MAGIC = 0x1.8p52 - (double)SHIFT;
    const double t = fma(y8, 11.541560327111707259, MAGIC);         // 8 log2 e; low dword = k - SHIFT
    const double k = t - MAGIC;
    const double r = fma(k, -0x1.62e42fefa39efp-4, y8);              // fl(ln2 / 8)
    double p = 0x1.a02041015378fp-13;
    p = fma(p, r, 0x1.6c1d00cea5bf1p-10);
    p = fma(p, r, 0x1.111111080fc42p-7);
    p = fma(p, r, 0x1.5555554653263p-5);
    p = fma(p, r, 0x1.5555555555689p-3);
    p = fma(p, r, 0x1.0000000000171p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    p = p * p;
    p = p * p;                                                       // exp(r)^4
#if SDPCUT_EXP_SCALE_LDEXP
    // (r4) v_ldexp_f64 with the low dword of t as its integer operand: ONE instruction for the scaled copy where the integer add
    // into the exponent field needs two (v_lshl_add_u32 on the high dword + a v_mov_b32 of the low one to complete the register
    // pair -- the unscaled p is still needed).  Same bits: both are exact scalings by 2^(k - SHIFT) in this range.
    const double ps = ldexp(p, __double2loint(t));
#else
    const double ps = __hiloint2double(__double2hiint(p) + (__double2loint(t) << 20), __double2loint(p));
#endif
    return fma(ps, p, addend);
}
__device__ __forceinline__ double exp_y8(double y8_in, double y8max)     // exp(8 y8_in), y8_in = -n / 4
{
    return exp_y8_scaled<0>(y8_in, y8max);
}
__device__ __forceinline__ double exp_m2n(double n, double y8max) { return exp_y8(n * -0.25, y8max); }

// The reciprocal is v_rcp_f64 (4.5e-8) + one cubically convergent step.  Absolute error of
// tansig vs the exact formula <= 1e-15.
__device__ __forceinline__ double tansig(double n)
{
    const double d = exp_m2n(n, 88.0) + 1.0;
    double q = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, q, 1.0);
    q = fma(q, fma(e, e, e), q);                        // q (1 + e + e^2)
    return fma(2.0, q, -1.0);
}

template <bool CLAMP = true>
__device__ __forceinline__ double tansig_y8(double y8)  // the accumulators of the MFMA kernel hold y/8 = -n/4 (NetDev::bias_q)
{
    const double d = CLAMP ? exp_y8_scaled<0, true>(y8, 88.0) + 1.0 : exp_y8_plus_bounded<0>(y8, 1.0);
    double q = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, q, 1.0);
    q = fma(q, fma(e, e, e), q);
    return fma(2.0, q, -1.0);
}

// HALF the denominator of tansig, h = (exp(-2n) + 1) / 2, so that tansig = 1/h - 1 (the factor 2
// of the formula is absorbed exactly by the exponent shift of exp_y8_scaled<1>).
template <bool CLAMP = true>
__device__ __forceinline__ double tansig_hden_y8(double y8, double y8max)
{
    if constexpr (CLAMP) return exp_y8_scaled<1, true>(y8, y8max) + 0.5;
    else return exp_y8_plus_bounded<1>(y8, 0.5);
}

// Four tansig values with ONE reciprocal: 1/h_i = (1 / (h0 h1 h2 h3)) * prod_{j != i} h_j.
// v_rcp_f64 plus its refinement is 6 issue slots (the kernel is bound by the VALU/MFMA
// instruction count): shared by four values the reciprocal part costs 3.5 slots per value
// instead of 7, and the halved denominators drop the doubling.  Every h_i is >= 1/2 and
// <= (1 + e^176) / 2 (y = -2n clamped to 176: tansig(-88) is -1 to 2e-76 either way), so the
// product neither underflows nor overflows; the extra roundings stay below 5e-16 relative.
// The four values are four neurons of ONE candidate (the rows of a C/D fragment), so a
// candidate's score does not depend on its neighbours in the wave.  (Sharing over the eight
// values of the two column tiles saved another 0.7 % but made duplicates of a candidate differ
// in the last bit depending on their position -- ties would no longer break by index.)
template <bool CLAMP = true>
__device__ __forceinline__ void tansig4(double &v0, double &v1, double &v2, double &v3)
{
    const double d0 = tansig_hden_y8<CLAMP>(v0, 22.0), d1 = tansig_hden_y8<CLAMP>(v1, 22.0);
    const double d2 = tansig_hden_y8<CLAMP>(v2, 22.0), d3 = tansig_hden_y8<CLAMP>(v3, 22.0);
    const double d01 = d0 * d1, d23 = d2 * d3;
    const double dd = d01 * d23;
    double q = __builtin_amdgcn_rcp(dd);
    const double e = fma(-dd, q, 1.0);
    q = fma(q, fma(e, e, e), q);
    const double q01 = q * d23, q23 = q * d01;          // 1/(h0 h1), 1/(h2 h3)
    v0 = fma(q01, d1, -1.0);
    v1 = fma(q01, d0, -1.0);
    v2 = fma(q23, d3, -1.0);
    v3 = fma(q23, d2, -1.0);
}

// tansig of one MFMA C/D fragment (rows 16 t + 4 r + q, r = 0..3); rows >= H are padding -> 0
template <int H, bool CLAMP = true>
__device__ __forceinline__ d4 tansig_tile(d4 c, int t)
{
    double v0 = c[0], v1 = c[1], v2 = c[2], v3 = c[3];
#ifndef SDPCUT_ABL_NOTANSIG     // tools/build_ablation.sh: timing experiments only
    tansig4<CLAMP>(v0, v1, v2, v3);
#endif
    d4 out;
    out[0] = (16 * t + 0 < H) ? v0 : 0.0;
    out[1] = (16 * t + 4 < H) ? v1 : 0.0;
    out[2] = (16 * t + 8 < H) ? v2 : 0.0;
    out[3] = (16 * t + 12 < H) ? v3 : 0.0;
    return out;
}

// The LDS strips of the MFMA kernel (feat, ynn) are private to one wave.  A wave's LDS
// instructions are issued and serviced in program order, so a write followed by a read of
// another lane's slot needs no workgroup barrier -- only that the compiler keeps the order and
// that the data has returned (lgkmcnt) before use.  Dropping __syncthreads() decouples the four
// waves of a workgroup: none waits for the slowest (PMC: SQ_WAIT_ANY 45 % of wave cycles).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v + v(lane ^ 32) and v + v(lane ^ 16) through v_permlane{32,16}_swap_b32 (gfx950) instead of ds_bpermute_b32: the swap of
// a register pair holding the same value leaves one register with the lower / even rows' values everywhere and the other with the
// upper / odd rows', and their sum is the butterfly sum on every lane -- bit for bit what v + __shfl_xor(v, 32 | 16) gives
// (addition commutes).  No LDS crossbar round trip (two dependent ones per reduction, ~250 cycles, at every layer boundary of a pass).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#ifndef SDPCUT_PERMLANE_SWAP
#define SDPCUT_PERMLANE_SWAP 1
#endif
__device__ __forceinline__ double xor_add32(double v)
{
#if SDPCUT_PERMLANE_SWAP
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u32x2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const u32x2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
#else
    return v + __shfl_xor(v, 32);
#endif
}
__device__ __forceinline__ double xor_add16(double v)
{
#if SDPCUT_PERMLANE_SWAP
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u32x2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u32x2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
#else
    return v + __shfl_xor(v, 16);
#endif
}
// v(lane ^ 32)
__device__ __forceinline__ double xor_get32(double v, int lane)
{
#if SDPCUT_PERMLANE_SWAP
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u32x2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);      // [0]: lower half everywhere, [1]: upper half everywhere
    const u32x2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double lower = __hiloint2double((int)b[0], (int)a[0]), upper = __hiloint2double((int)b[1], (int)a[1]);
    return (lane & 32) ? lower : upper;
#else
    return __shfl_xor(v, 32);
#endif
}

// Tail rows of a hidden layer on the VALU: ts[u] holds this lane's partial dot product of tail
// neuron u over the k-slots it owns (n = 4 s + q); the four k-slot lanes of a candidate column
// (lane, lane^16, lane^32, lane^48) are summed, and lane q keeps neuron u = q in register 0 of
// the last row tile -- exactly where the MFMA C/D layout would have put it.
// Both column tiles of a pass share ONE tansig evaluation: after the two xor-adds every lane of a
// column holds the full sums, so lanes q = 0, 1 take tile j = 0 and lanes q = 2, 3 tile j = 1
// (neuron u = q & 1); a final xor-32 shuffle hands the j = 1 values to lanes q = 0, 1.
template <int NT, bool CLAMP = true>
__device__ __forceinline__ void tail_rows2(const double (&ts)[2][NT ? NT : 1], const double *bias, int q, d4 &out0,
                                           d4 &out1)
{
    static_assert(NT <= 2, "at most two tail neurons");
    double pre = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            double v = ts[j][u];
            v = xor_add16(v);
            v = xor_add32(v);
            pre = (q == 2 * j + u) ? v + bias[u] : pre;
        }
    const double t = tansig_y8<CLAMP>(pre);
    const double t1 = xor_get32(t, 16 * q);
    out0 = d4{0.0, 0.0, 0.0, 0.0};
    out1 = d4{0.0, 0.0, 0.0, 0.0};
    out0[0] = (q < NT) ? t : 0.0;
    out1[0] = (q < NT) ? t1 : 0.0;
}

// The same for a pass of ONE column tile (the three-waves-per-SIMD variant of the kernel): lanes q < NT evaluate the tansig
// of neuron u = q, the other k-slot lanes idle through it.
template <int NT, bool CLAMP = true>
__device__ __forceinline__ void tail_rows1(const double (&ts)[1][NT ? NT : 1], const double *bias, int q, d4 &out0)
{
    static_assert(NT <= 2, "at most two tail neurons");
    double pre = 0.0;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        double v = ts[0][u];
        v = xor_add16(v);
        v = xor_add32(v);
        pre = (q == u) ? v + bias[u] : pre;
    }
    const double t = tansig_y8<CLAMP>(pre);
    out0 = d4{0.0, 0.0, 0.0, 0.0};
    out0[0] = (q < NT) ? t : 0.0;
}

// Timing experiments (tools/build_ablation.sh; results are wrong by design): drop the bias or
// weight-fragment loads to see what their latency costs.
// Biases, tail-row weights and output weights (<= 8.5 KB) are copied to LDS once per workgroup:
// they are read at every tile / layer boundary, exactly where a wave has nothing else in flight
// to cover an L2 round trip (ds_read ~100 cycles instead of ~700).
#ifndef SDPCUT_SMALL_IN_LDS
#define SDPCUT_SMALL_IN_LDS 1
#endif
// (The A-fragments themselves were tried in LDS too -- 44.5 KB for the 3-variable net, two
// workgroups per CU still fit: no gain, the register ring already hides their L2 latency.)
#ifdef SDPCUT_ABL_NOBIAS
#define BIAS_AT(i) (0.125 + 0.0 * (double)(i))
#elif SDPCUT_SMALL_IN_LDS
#define BIAS_AT(i) s_bias[i]
#else
#define BIAS_AT(i) net.bias_q[i]
#endif
#if SDPCUT_SMALL_IN_LDS
#define WTAIL_AT(i) s_wtail[i]
#define WOUT_AT(i) s_wout[i]
#define BIAS_PTR s_bias
#else
#define WTAIL_AT(i) net.wtail[i]
#define WOUT_AT(i) net.wout[i]
#define BIAS_PTR net.bias_q
#endif
#ifdef SDPCUT_ABL_NOWLOAD
#define WFRAG_AT(i) (0.01 * (double)((i) & 7))
#else
#define WFRAG_AT(i) wf[i]
#endif

// Timing experiment (tools/build_ablation.sh ...:PHASETIME): cycles a wave spends between the phase
// boundaries of a tile, printed by a few waves.  [0-1] gather wait, [1-2] Jacobi, [2-3] staging,
// [3-4] the MLP passes.
#ifdef SDPCUT_ABL_PHASETIME
#define PHASE_DECL unsigned long long ph_t[5] = {0, 0, 0, 0, 0}, ph_acc[4] = {0, 0, 0, 0}; int ph_n = 0
#define PHASE_MARK(i)                                                      \
    do {                                                                   \
        ph_t[i] = __builtin_readcyclecounter();                            \
        if ((i) > 0) ph_acc[(i) - 1] += ph_t[i] - ph_t[(i) - 1];           \
        if ((i) == 4) ++ph_n;                                              \
    } while (0)
#define PHASE_WAITMEM asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PHASE_REPORT                                                                                        \
    if ((threadIdx.x & 63) == 0 && (blockIdx.x % 509) == 3)                                                 \
    printf("blk %d wave %d tiles %d: gather %llu jacobi %llu stage %llu mlp %llu cycles/tile\n", (int)blockIdx.x, \
           (int)(threadIdx.x >> 6), ph_n, ph_acc[0] / (ph_n ? ph_n : 1), ph_acc[1] / (ph_n ? ph_n : 1),      \
           ph_acc[2] / (ph_n ? ph_n : 1), ph_acc[3] / (ph_n ? ph_n : 1))
#elif defined(SDPCUT_ABL_CLOCKS)
// shader clock actually sustained while the kernel runs: core-clock counter against the 100 MHz one
#define PHASE_DECL const unsigned long long ph_c0 = clock64(), ph_w0 = wall_clock64()
#define PHASE_MARK(i)
#define PHASE_WAITMEM
#define PHASE_REPORT                                                                                         \
    if (threadIdx.x == 0 && (blockIdx.x % 509) == 3) {                                                       \
        const unsigned long long dc = clock64() - ph_c0, dw = wall_clock64() - ph_w0;                        \
        printf("blk %d: %llu core cycles in %llu ticks of 10 ns -> %.0f MHz\n", (int)blockIdx.x, dc, dw,     \
               dw ? 100.0 * (double)dc / (double)dw : 0.0);                                                  \
    }
#else
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_WAITMEM
#define PHASE_REPORT
#endif

// ------------------------------------------------------------------------------------------
// MFMA kernel.  K candidate size, H hidden width, NH hidden layers; FUSE = TK_MODE_FEAS / OPT / STRONG:
// also count the class members by the leading radix digit of that mode's selection keys (ScoreArgs::tk).
// CLAMP = false (NetDev::unclamped_ok): the tansig clamps are dropped and the staged inputs are
// clamped to [-3, 3] instead (inactive for every x in [0, 1], |q| <= 1/k, see sdpcut_set_network).
// J = 16-candidate column tiles per pass.  2 (default): every weight fragment feeds two MFMAs, ~230 registers, two waves
// per SIMD.  1: half the accumulator state (<= 168 registers: three waves per SIMD), twice the fragment loads and tail
// evaluations per candidate -- the experiment of VERDICT r2 item 4, measured in DESIGN.md section 5.
#ifndef SDPCUT_MFMA_J
#define SDPCUT_MFMA_J 2
#endif
#ifndef SDPCUT_MFMA_J_K2
#define SDPCUT_MFMA_J_K2 SDPCUT_MFMA_J
#endif
#ifndef SDPCUT_MFMA_J_K3
#define SDPCUT_MFMA_J_K3 SDPCUT_MFMA_J
#endif
#ifndef SDPCUT_MFMA_J_K4
#define SDPCUT_MFMA_J_K4 SDPCUT_MFMA_J
#endif
#ifndef SDPCUT_MFMA_J_K5
#define SDPCUT_MFMA_J_K5 SDPCUT_MFMA_J
#endif
#ifndef SDPCUT_MFMA_J1_WAVES
#define SDPCUT_MFMA_J1_WAVES 3
#endif
constexpr int mfma_cols(int K) { return K == 2 ? SDPCUT_MFMA_J_K2 : K == 3 ? SDPCUT_MFMA_J_K3 : K == 4 ? SDPCUT_MFMA_J_K4 : SDPCUT_MFMA_J_K5; }

// LDS of one workgroup of the MFMA kernel for size class K (a union of these serves the launch over all classes)
#ifndef SDPCUT_PF_SCORE_KMASK
#define SDPCUT_PF_SCORE_KMASK (1 << 3)      // candidate sizes whose score kernel counts the fine histogram (bit k; 0: none; see score_mfma_body)
#endif
constexpr bool pf_score_k(int k) { return ((SDPCUT_PF_SCORE_KMASK >> k) & 1) != 0; }
template <int K, int H, int NH>
struct MfmaLds {
    static constexpr int S0 = (K + K * (K + 1) / 2 + 3) / 4;
    static constexpr int T = (H + 15) / 16;
    static constexpr int NT = (H - 16 * (T - 1) <= 4) ? H - 16 * (T - 1) : 0;
    double feat[4][S0 * 4][64];  // per wave: feature-major strip of 64 candidates
    double ynn[4][64];           // per wave: raw network outputs
    double s_bias[NH * 64];
    double s_wtail[NT ? NH * 4 * 64 : 1];
    double s_wout[64];
    uint32_t tk_hist[256];       // leading-digit histogram of the selection that follows (A.tk != nullptr)
    uint32_t tk_cnt[2];
    uint32_t s_strong;
    uint32_t pf_tab[pf_score_k(K) ? PF_BINS / 2 : 1];    // (r5) the class members by window code, 16-bit counters, two per word (topk_dev.h)
};

// bid / nblk: this workgroup's index among the nblk workgroups that serve the class (blockIdx.x / gridDim.x of a launch over
// one class; the launch over all classes of a list hands every class its own range of workgroups, score_mfma_all_kernel)
template <int K, int H, int NH, int FUSE, bool CLAMP, int JK>
__device__ __forceinline__ void score_mfma_body(const ScoreArgs A, MfmaLds<K, H, NH> &S, const int bid, const int nblk)
{
    constexpr int M = K * (K + 1) / 2;
    constexpr int DIN = K + M;
    constexpr int S0 = (DIN + 3) / 4;      // k-steps of the input layer
    constexpr int SH = (H + 3) / 4;        // k-steps of a hidden->hidden layer
    constexpr int T = (H + 15) / 16;       // 16-neuron row tiles
    // A last tile with <= 4 live rows (H = 50: neurons 48, 49) would cost a full 16-row MFMA
    // per k-step for 1/8 of the work: those rows run on the VALU instead (tail_rows below).
    constexpr int NT = (H - 16 * (T - 1) <= 4) ? H - 16 * (T - 1) : 0;
    constexpr int TM = NT ? T - 1 : T;     // row tiles computed with MFMA
    static_assert(JK == 1 || JK == 2, "one or two column tiles per pass");
    static_assert(T == 4, "hidden width must be in 49..64");

    auto &feat = S.feat;
    auto &ynn = S.ynn;

    const int lane = threadIdx.x & 63;
    // (wave-uniform by construction; said so to the compiler: the wave's range, its strip loop and pass counts are then scalar)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q = lane >> 4;      // MFMA k-slot / output row group
    const int c16 = lane & 15;    // MFMA column (candidate within a 16-tile)
    const NetDev &net = A.net;
    // Work split (r3).  The list is cut into STRIPS of A.strip candidates, and wave g of the launch takes strips g, g + W, g + 2W ...
    // (W waves in the launch: the four waves of a workgroup take four consecutive strips, the workgroups move through the list
    // together -- at 10^8 candidates that keeps the 3 GB of index sets and scores the resident waves touch close together).
    //  * Long lists: strips of 64 candidates = two passes of the MLP over two 16-candidate column tiles each, eight workgroups per
    //    CU (short-lived: the dispatcher balances them).
    //  * Lists that leave the device part-empty (<= 32 candidates per resident wave, i.e. <= 65 536 -- most real covers): strips
    //    of 32, ONE pass per wave.  A launch over such a list takes as long as its slowest wave -- phase A plus its passes, 45-95 us
    //    of dependent stages -- and twice as many waves with one pass each finish sooner than half as many with two.  (Finer does
    //    not pay: a single-tile pass of a lone wave takes as long as a two-tile pass, profiles/r03_k3_kernel_time_vs_list_length.txt.)
    // The last strip of a list may hold fewer candidates: it runs the passes its column tiles need, the last one over a single
    // tile if their number is odd (mlp_pass<1>: the same arithmetic per candidate, bit-equal scores).
    // (r5) A.spread (lists of ONE strip per wave -- most real covers): wave w of workgroup b is wave w * nblk + b of the launch, the
    // four waves of a workgroup work on four DISTANT quarters of the list.  Real covers are enumerated index set by index set,
    // neighbours share variables and scores: 256 consecutive candidates rich in members of the head would otherwise be one
    // workgroup's whole share, which then reports fewer of them than it holds and sends the selection through its radix passes
    // (pf_retire_table, topk_dev.h; 13 of 191 recorded rounds).  Longer lists keep the consecutive strips (a workgroup's share
    // already comes from several rounds of the list; spreading its waves cost the 10^6-candidate launch 3 us).  Scores do not
    // depend on who computes them.
    const int64_t gw = A.spread ? (int64_t)wave * nblk + bid : (int64_t)bid * 4 + wave;
    const int64_t wstride = (int64_t)nblk * 4 * A.strip;      // candidates between two strips of one wave
    const int64_t c_first = gw * A.strip;
    // ... and the LAST round of a list of a few rounds, when it is nearly full (r4): 10^6 candidates are 7.63 strips per resident
    // wave slot, every slot ran 8 -- 4.6 % of the kernel idle at its end.  Now the round-robin part ends at the last FULL round
    // (rr_end) and what is left is split evenly in column tiles of 16: three or four per wave (a three-tile strip = one two-tile
    // pass + one single-tile pass), every workgroup the same for its four waves.
    const int64_t t_tiles = gw < A.tail_nhi ? A.tail_hi : A.tail_lo;
    const int64_t t_start0 = A.rr_end + 16 * (gw < A.tail_nhi ? gw * A.tail_hi : A.tail_nhi * A.tail_hi + (gw - A.tail_nhi) * A.tail_lo);
    const int64_t t_start = t_start0 < A.n ? t_start0 : A.n;
    const int64_t t_end = t_start + 16 * t_tiles < A.n ? t_start + 16 * t_tiles : A.n;      // (empty when t_start == t_end)
    bool tail = c_first >= A.rr_end;                  // this wave's current strip is its tail strip
    int64_t s0 = tail ? t_start : c_first;
    bool more = tail ? t_start < t_end : true;

    // The index set (and the output slot) of the NEXT strip are requested before phase B of the
    // current one: the first of the two dependent memory round trips of phase A (HBM: indices, then
    // L2: the gathers they address) is off the critical path.
    // (The first strip's request goes out before the LDS preload below so that the two latencies of a
    // workgroup's start overlap.)
    int32_t s_nxt[K];
    int32_t orig_nxt = 0;
    if (more) {
        const int64_t lim0 = tail ? t_end : (s0 + A.strip < A.rr_end ? s0 + A.strip : A.rr_end);
        const int64_t c0 = s0 + lane;
        const int64_t cc0 = c0 < lim0 ? c0 : s0;
        load_index_set<K>(s_nxt, A.set, A.n, cc0);
        orig_nxt = A.orig[cc0];
    }

#if SDPCUT_SMALL_IN_LDS
    auto &s_bias = S.s_bias;
    auto &s_wtail = S.s_wtail;
    auto &s_wout = S.s_wout;
    if (A.flags & SDPCUT_NN) {     // uniform; an eigenvalue-only launch may come without a network
        for (int i = threadIdx.x; i < NH * 64; i += 256) s_bias[i] = net.bias_q[i];
        if constexpr (NT > 0)
            for (int i = threadIdx.x; i < NH * 4 * 64; i += 256) s_wtail[i] = net.wtail[i];
        if (threadIdx.x < 64) s_wout[threadIdx.x] = net.wout[threadIdx.x];
        __syncthreads();
    }
#endif

    // leading-digit histograms of the selection that follows (A.tk != nullptr)
    auto &tk_hist = S.tk_hist;
    auto &tk_cnt = S.tk_cnt;
    uint32_t c_viol = 0, c_pos = 0, c_strong = 0;     // per lane (vector registers: the scalar file is full)
    // (r5) the fine histogram of the selection's class (topk_dev.h) is compiled into the kernel of 3-variable candidates only:
    // merely present -- not executed -- it costs the 4-variable kernel 11 us on the 1.7e6-candidate cover of spar125-075-1 (240
    // registers, 112 bytes of scratch: the allocation of its hot loop moves), executed 22, against the 12 us the selection saves
    // (profiles/r05_fine_histogram_score_kernel_variants.txt); the 2-variable kernel loses 7 us on 10^6 candidates the same way
    // (profiles/r05_vs_r4_same_box.txt); on 10^6 three-variable candidates it costs 5-6 and saves 10.  Feasibility rounds -- three
    // quarters of a BoxQP run -- count in the eigenvalue kernel (eig.hip) for every size.
    constexpr bool PF = FUSE != 0 && pf_score_k(K);
    auto &pf_tab = S.pf_tab;
    if constexpr (FUSE != 0) {
        tk_hist[threadIdx.x] = 0;
        if (threadIdx.x < 2) tk_cnt[threadIdx.x] = 0;
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < PF_BINS / 2 / 256; ++j) pf_tab[threadIdx.x + 256 * j] = 0;
        }
        __syncthreads();
    }

    PHASE_DECL;
    while (more) {
        PHASE_MARK(0);
        const int64_t lim = tail ? t_end : (s0 + A.strip < A.rr_end ? s0 + A.strip : A.rr_end);      // one past this strip's last candidate
        // the strip after this one: the next round-robin strip, or the tail strip behind the last of them
        const bool nx_rr = !tail && s0 + wstride < A.rr_end;
        const bool nx_tail = !tail && !nx_rr;
        const int64_t nx_s0 = nx_rr ? s0 + wstride : t_start;
        const int64_t nx_lim = nx_rr ? (nx_s0 + A.strip < A.rr_end ? nx_s0 + A.strip : A.rr_end) : t_end;
        const bool nx_more = nx_rr || (nx_tail && t_start < t_end);
        const int64_t c = s0 + lane;
        const bool valid = c < lim;
        int32_t s_cur[K];
#pragma unroll
        for (int a = 0; a < K; ++a) s_cur[a] = s_nxt[a];
        const int32_t out_idx = orig_nxt;
        Cand<K> cd;
        gather_candidate<K>(cd, s_cur, A.vars, A.Q, A.nv, A.L, (A.flags & SDPCUT_NN) != 0);
        if (nx_more) {                       // uniform per wave
            const int64_t c1 = nx_s0 + lane;
            const int64_t cc1 = c1 < nx_lim ? c1 : nx_s0;
            load_index_set<K>(s_nxt, A.set, A.n, cc1);
            orig_nxt = A.orig[cc1];
        }

        double lam = 0.0;
        PHASE_WAITMEM;
        PHASE_MARK(1);
        if (A.flags & SDPCUT_EIG) {
            lam = candidate_eigmin<K>(cd, s_cur, A.vars, A.nv, A.L);
            if (valid) A.eig_out[out_idx] = lam;
        }
        PHASE_MARK(2);
        if (!(A.flags & SDPCUT_NN)) {           // uniform branch
            if constexpr (FUSE != 0) {
                const bool viol = valid && lam < SDPCUT_NEG_EIGVAL;      // (only TK_MODE_FEAS ranks without the network)
                const uint64_t key = key_of(-lam);
                hist_add_few(tk_hist, (uint32_t)(key >> 56), viol);
                if (PF && viol) { const int f = pf_code(key, true); atomicAdd(&pf_tab[f >> 1], (f & 1) ? 0x10000u : 1u); }
                c_viol += viol;
            }
            tail = tail || nx_tail; s0 = nx_s0; more = nx_more;
            continue;
        }

        // ---- stage mapminmax'ed inputs (neural_net_3D.m:69-73): xp = (v - xoffset)*gain + ymin
#pragma unroll
        for (int i = 0; i < S0 * 4; ++i) {
            double xp = 0.0;
            if (i < DIN) {
                const double v = (i < K) ? cd.x[i < K ? i : 0] : cd.q[i >= K ? i - K : 0];
                xp = (v - net.inmap[i]) * net.inmap[DIN + i] + net.ymin;
                if constexpr (!CLAMP) xp = min_f64_raw(max_f64_raw(xp, -SDPCUT_INPUT_CLAMP), SDPCUT_INPUT_CLAMP);
            }
            feat[wave][i][lane] = xp;
        }
        wave_lds_sync();
        PHASE_MARK(3);

        // Weight fragments requested one stage ahead of their use (SDPCUT_XPREFETCH): the input layer's
        // tile t+1 while tile t computes, the first fragments of a hidden layer before the last tansig of
        // the layer in front of it, the first input tile of the next pass before the output layer.
        // Measured: the 2-variable kernel gains 1.5 % (0.399 -> 0.393 ms), the 3-variable one loses 0.6 %
        // (its other wave already covers the L2 round trips at the stage boundaries; 5 more live
        // registers cost more), the 4/5-variable ones have no registers left (spills): on for K = 2 only.
#ifndef SDPCUT_XPREFETCH
#define SDPCUT_XPREFETCH 1
#endif
#ifndef SDPCUT_XPREFETCH_MAXK
#define SDPCUT_XPREFETCH_MAXK 2
#endif
// (r4) ... and for K = 5 again: with lambda_min by lmin.h instead of the 6x6 Jacobi the 5-variable kernel has the registers the
// prefetch needs -- 584 -> 566 us on 1e6 candidates (-3 %); K = 3 still loses 0.6 %, K = 4 is indifferent.
#ifndef SDPCUT_XPREFETCH_MINK
#define SDPCUT_XPREFETCH_MINK 5
#endif
#ifndef SDPCUT_RING_DEPTH
#define SDPCUT_RING_DEPTH 4
#endif
        constexpr int RD = SDPCUT_RING_DEPTH;
        constexpr bool XP = SDPCUT_XPREFETCH && (K <= SDPCUT_XPREFETCH_MAXK || K >= SDPCUT_XPREFETCH_MINK);
        double a_in[S0];              // input-layer fragments of the tile about to run
        double pre[RD];               // head of the next hidden layer's fragment stream
        if constexpr (XP) {
#pragma unroll
            for (int s = 0; s < S0; ++s) a_in[s] = net.wfrag[s * 64 + lane];
        }
        // one pass of the MLP over JJ column tiles (16 JJ candidates from column col0 of the wave's strip)
        auto mlp_pass = [&](auto jj_tag, const int col0) __attribute__((always_inline)) {
            constexpr int J = decltype(jj_tag)::value;
            // B fragments of the input layer: B[k = 4s + q][col = candidate]
            double bin[S0][J];
#pragma unroll
            for (int s = 0; s < S0; ++s)
#pragma unroll
                for (int j = 0; j < J; ++j) bin[s][j] = feat[wave][4 * s + q][col0 + 16 * j + c16];

            d4 prev[T][J], cur[T][J];
            const double *wf = net.wfrag;
            // ---------------- input layer
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                d4 bias;
#pragma unroll
                for (int r = 0; r < 4; ++r) bias[r] = BIAS_AT(16 * t + 4 * r + q);
#pragma unroll
                for (int j = 0; j < J; ++j) cur[t][j] = bias;
                double a_nx[S0];
                if constexpr (XP) {
                    if (t + 1 < TM) {
#pragma unroll
                        for (int s = 0; s < S0; ++s) a_nx[s] = WFRAG_AT(((t + 1) * S0 + s) * 64 + lane);
                    } else if (NH > 1) {      // the first hidden layer's stream starts behind the T input tiles
#pragma unroll
                        for (int g = 0; g < RD - 1; ++g) pre[g] = WFRAG_AT((T * S0 + g) * 64 + lane);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int s = 0; s < S0; ++s) {
                    const double a = XP ? a_in[s] : WFRAG_AT((t * S0 + s) * 64 + lane);
#pragma unroll
                    for (int j = 0; j < J; ++j)
                        cur[t][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bin[s][j], cur[t][j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < J; ++j) cur[t][j] = tansig_tile<H, CLAMP>(cur[t][j], t);
                __builtin_amdgcn_sched_barrier(0);
                if (XP && t + 1 < TM) {
#pragma unroll
                    for (int s = 0; s < S0; ++s) a_in[s] = a_nx[s];
                }
            }
            if constexpr (NT > 0) {
                double ts[J][NT ? NT : 1];
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int u = 0; u < NT; ++u) ts[j][u] = 0.0;
#pragma unroll
                for (int s = 0; s < S0; ++s)
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        const double w = WTAIL_AT(u * 64 + 4 * s + q);
#pragma unroll
                        for (int j = 0; j < J; ++j) ts[j][u] = fma(bin[s][j], w, ts[j][u]);
                    }
                if constexpr (J == 2) tail_rows2<NT, CLAMP>(ts, BIAS_PTR + 16 * (T - 1), q, cur[T - 1][0], cur[T - 1][J - 1]);
                else tail_rows1<NT, CLAMP>(ts, BIAS_PTR + 16 * (T - 1), q, cur[T - 1][0]);
            }
            wf += T * S0 * 64;
            // ---------------- hidden -> hidden layers (rolled: bounds code size and live ranges)
#pragma unroll 1
            for (int l = 1; l < NH; ++l) {
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int j = 0; j < J; ++j) prev[t][j] = cur[t][j];
                // The A-fragment stream of the layer is one contiguous sequence g = t*SH + s.  Left
                // alone hipcc loads each fragment right before its two MFMAs and waits vmcnt(0)
                // (an L2 round trip per 128 MFMA cycles, the dominant stall in the first PMC run):
                // a ring of RD fragments keeps RD-1 loads in flight; the sched_barriers pin the order.
                constexpr int NG = TM * SH;
                static_assert(NG >= RD - 1, "fragment stream shorter than the ring");
                double ring[RD];
#pragma unroll
                for (int g = 0; g < RD - 1 && g < NG; ++g) ring[g] = XP ? pre[g] : WFRAG_AT(g * 64 + lane);
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    d4 bias;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bias[r] = BIAS_AT(l * 64 + 16 * t + 4 * r + q);
#pragma unroll
                    for (int j = 0; j < J; ++j) cur[t][j] = bias;
#pragma unroll
                    for (int s = 0; s < SH; ++s) {
                        const int g = t * SH + s;
                        if (g + RD - 1 < NG) ring[(g + RD - 1) % RD] = WFRAG_AT((g + RD - 1) * 64 + lane);
                        if (XP && g == NG - 1) {
                            // the stage behind this layer: the next hidden layer's head, or -- behind the last
                            // one -- the first input tile of the next pass
                            if (l + 1 < NH) {
#pragma unroll
                                for (int gg = 0; gg < RD - 1; ++gg) pre[gg] = WFRAG_AT((T * SH + gg) * 64 + lane);
                            } else {
#pragma unroll
                                for (int ss = 0; ss < S0; ++ss) a_in[ss] = net.wfrag[ss * 64 + lane];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < J; ++j)
                            cur[t][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[g % RD], prev[s / 4][j][s % 4],
                                                                             cur[t][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) cur[t][j] = tansig_tile<H, CLAMP>(cur[t][j], t);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NT > 0) {
                    double ts[J][NT ? NT : 1];
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int u = 0; u < NT; ++u) ts[j][u] = 0.0;
#pragma unroll
                    for (int s = 0; s < SH; ++s)
#pragma unroll
                        for (int u = 0; u < NT; ++u) {
                            const double w = WTAIL_AT((l * 4 + u) * 64 + 4 * s + q);
#pragma unroll
                            for (int j = 0; j < J; ++j) ts[j][u] = fma(prev[s / 4][j][s % 4], w, ts[j][u]);
                        }
                    if constexpr (J == 2) tail_rows2<NT, CLAMP>(ts, BIAS_PTR + l * 64 + 16 * (T - 1), q, cur[T - 1][0], cur[T - 1][J - 1]);
                    else tail_rows1<NT, CLAMP>(ts, BIAS_PTR + l * 64 + 16 * (T - 1), q, cur[T - 1][0]);
                }
                wf += T * SH * 64;
            }
            // ---------------- linear output layer: dot over this lane's 16 neurons, then the
            // four k-slot lanes (q = 0..3) of each candidate column are summed by shuffles
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double part = 0.0;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * t + 4 * r < H) part = fma(cur[t][j][r], WOUT_AT(16 * t + 4 * r + q), part);
                part = xor_add16(part);
                part = xor_add32(part);
                if (q == 0) ynn[wave][col0 + 16 * j + c16] = part;
            }
        };
        {
            const int units = (int)((lim - s0 + 15) >> 4);      // column tiles of this strip that hold candidates
            // (a whole strip has its own loop with a constant trip count, and the rare paths are marked so: with one generic loop
            // the 5-variable kernel ran 1.5 % slower than before the split, this way 0.8 %, the 3- and 4-variable ones 1 % faster)
            if (__builtin_expect(units == 4, 1)) {
#pragma unroll 1
                for (int pass = 0; pass < 4 / JK; ++pass) mlp_pass(std::integral_constant<int, JK>{}, 16 * JK * pass);
            } else {
#pragma unroll 1
                for (int pass = 0; pass < units / JK; ++pass) mlp_pass(std::integral_constant<int, JK>{}, 16 * JK * pass);
                if constexpr (JK == 2) {
                    if (units & 1) mlp_pass(std::integral_constant<int, 1>{}, 16 * (units - 1));
                }
            }
        }
        wave_lds_sync();
        {
            // neural_net_3D.m:60-62, 81-85: y = (a + b - ymin)/gain + xoffset;  then :582
#pragma clang fp contract(off)
            double acc = ynn[wave][lane];
            acc = acc + net.b_out;
            const double y = (acc - net.y_ymin) / net.y_gain + net.y_xoffset;
            double obj = cd.negSM;
            obj = obj + y * cd.max_elem;
            if (valid) A.obj_out[out_idx] = obj;
            const bool viol = valid && (A.flags & SDPCUT_EIG) && lam < SDPCUT_NEG_EIGVAL, pos = valid && obj > 0.0;
            c_strong += viol && pos;
            if constexpr (FUSE != 0) {
                c_viol += viol;
                c_pos += pos;
                const bool member = FUSE == TK_MODE_OPT ? valid : FUSE == TK_MODE_FEAS ? viol : (viol && pos);
                const uint64_t key = key_of(FUSE == TK_MODE_FEAS ? -lam : obj);
                hist_add_few(tk_hist, (uint32_t)(key >> 56), member);
                if (PF && member) { const int f = pf_code(key, FUSE == TK_MODE_FEAS); atomicAdd(&pf_tab[f >> 1], (f & 1) ? 0x10000u : 1u); }      // (LDS, no return value: one ds_add per candidate)
            }
        }
        wave_lds_sync();   // feat / ynn are rewritten by the next tile
        PHASE_MARK(4);
        tail = tail || nx_tail; s0 = nx_s0; more = nx_more;
    }
    PHASE_REPORT;
    if (A.strong_out) {      // uniform: one no-return atomic per workgroup, into one of eight replicas
        uint32_t &s_strong = S.s_strong;
        if (threadIdx.x == 0) s_strong = 0;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) c_strong += __shfl_xor((int)c_strong, off);
        if (lane == 0 && c_strong) atomicAdd(&s_strong, c_strong);
        __syncthreads();
        if (threadIdx.x == 0 && s_strong)
            __hip_atomic_fetch_add((unsigned long long *)&A.strong_out[blockIdx.x & 7], (unsigned long long)s_strong,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (FUSE != 0) {
        // No ticket, nobody waits: the kernel boundary orders the atomics before the selection.  (A ticket per
        // workgroup -- the last one resolving the digit, as the selection's own passes do -- costs a drain
        // of the workgroup's stores plus an atomic round trip before each of the 2048 workgroups may
        // retire: +14 us on this kernel.)
        for (int off = 32; off > 0; off >>= 1) {
            c_viol += __shfl_xor((int)c_viol, off);
            c_pos += __shfl_xor((int)c_pos, off);
        }
        if (lane == 0) {
            if (c_viol) atomicAdd(&tk_cnt[0], c_viol);
            if (c_pos) atomicAdd(&tk_cnt[1], c_pos);
        }
        __syncthreads();
        if (threadIdx.x < 2 && tk_cnt[threadIdx.x])
            __hip_atomic_fetch_add((unsigned long long *)(threadIdx.x ? &A.tk->pos_rep[blockIdx.x % TK_SHREP] : &A.tk->viol_rep[blockIdx.x % TK_SHREP]),
                                   (unsigned long long)tk_cnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk_hist[threadIdx.x])
            __hip_atomic_fetch_add(&A.tk->hist_score[blockIdx.x % TK_SHREP][threadIdx.x], tk_hist[threadIdx.x], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (PF) {
            if (A.pf_mloc > 0) pf_retire_table(A.tk, pf_tab, A.pf_mloc);
        }      // (behind the barrier above: the table is complete)
    }
}

template <int K, int H, int NH, int FUSE = 0, bool CLAMP = true, int JK = 2>
__global__ __launch_bounds__(256, (JK == 1 ? SDPCUT_MFMA_J1_WAVES : 2)) void score_mfma_kernel(ScoreArgs A)
{
    __shared__ MfmaLds<K, H, NH> S;
    score_mfma_body<K, H, NH, FUSE, CLAMP, JK>(A, S, (int)blockIdx.x, (int)gridDim.x);
}

// ONE launch for every size class of a list (r3).  Real covers hold one large class and a few sets of the smaller sizes
// (spar100-050-1, dim 5: 72 673 five-variable sets, 103 of four, 1 of three); a launch per class costs what one pass costs however
// few candidates it holds, and side streams run side by side only if the process's streams were handed different hardware queues.
// Here every class gets its own range of workgroups of one launch, the largest class first; a workgroup serves exactly one
// class (its code is the single-class kernel's, its LDS a union of the classes').
struct ScoreArgsAll {
    ScoreArgs a[SDPCUT_MAX_K - 1];      // classes in launch order
    int32_t k[SDPCUT_MAX_K - 1];        // their sizes
    int32_t bend[SDPCUT_MAX_K - 1];     // one past their last workgroup
    int32_t nclasses;
};

template <int FUSE, bool CLAMP>
__global__ __launch_bounds__(256, 2) void score_mfma_all_kernel(ScoreArgsAll AA)
{
    __shared__ union LdsAll {
        MfmaLds<2, 64, 3> l2;
        MfmaLds<3, 50, 3> l3;
        MfmaLds<4, 50, 3> l4;
        MfmaLds<5, 64, 4> l5;
        __device__ LdsAll() {}
    } S;
    int c = 0;
    while (c + 1 < AA.nclasses && (int)blockIdx.x >= AA.bend[c]) ++c;      // uniform
    const int b0 = c ? AA.bend[c - 1] : 0;
    const int bid = (int)blockIdx.x - b0, nblk = AA.bend[c] - b0;
    switch (AA.k[c]) {
    case 2: score_mfma_body<2, 64, 3, FUSE, CLAMP, 2>(AA.a[c], S.l2, bid, nblk); break;
    case 3: score_mfma_body<3, 50, 3, FUSE, CLAMP, 2>(AA.a[c], S.l3, bid, nblk); break;
    case 4: score_mfma_body<4, 50, 3, FUSE, CLAMP, 2>(AA.a[c], S.l4, bid, nblk); break;
    default: score_mfma_body<5, 64, 4, FUSE, CLAMP, 2>(AA.a[c], S.l5, bid, nblk); break;
    }
}

// ------------------------------------------------------------------------------------------
// VALU kernel: lane = candidate, activations in registers, weights as SCALAR operands.
//
// On gfx950 v_mfma_f64_16x16x4_f64 and v_fma_f64 share the fp64 datapath: they do not overlap
// (profiles/r01_ubench_mfma_valu_overlap.txt: MFMA-only 0.85 ms, FMA-only 0.94 ms, both on one
// SIMD 1.81 ms) and peak at the same 78.6 TFLOP/s.  The MFMA form pads 50 neurons to 64 rows
// (26 % wasted FLOPs); here every fp64 FMA is a useful one.  Weights are wave-uniform, so they
// are read through the scalar cache (s_load_dwordx16 = 8 weights) and enter v_fma_f64 as SGPR
// operands: no LDS, no vector memory traffic in the MLP at all.  Eight output neurons are
// accumulated at once (8 independent FMA chains hide the fp64 latency); weights are packed
// host-side as [layer][j/8][i][j%8] so that each (block, i) is one 64-byte scalar load.
typedef const __attribute__((address_space(4))) double *cdouble_p;   // constant AS => SMEM loads

// Scalar loads return out of order, so the only usable wait is lgkmcnt(0): the weight stream is
// software-pipelined in batches of two input steps (2 x s_load_dwordx16 = 16 weights): batch
// g+1 is issued, then the 16 FMAs of batch g run while it is in flight.  The sched_barriers pin
// that order (left alone, hipcc issues each load right in front of its first use and eats the
// full scalar-cache latency every 8 FMAs).
template <int FAN, int H>
__device__ __forceinline__ void dense_tansig(cdouble_p wv, cdouble_p bias, const double (&in)[FAN], double (&out)[H])
{
    constexpr int JB = 8, NB = (H + JB - 1) / JB;
    constexpr int NBAT = (FAN + 1) / 2;          // batches of two input steps per output block
    double wa[2 * JB], wb[2 * JB];            // the two weight buffers (SGPRs)
#pragma unroll
    for (int t = 0; t < 2 * JB; ++t) wa[t] = wv[t];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        double acc[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) acc[jj] = bias[jb * JB + jj];
#pragma unroll
        for (int bt = 0; bt < NBAT; ++bt) {
            const int g = jb * NBAT + bt;             // global batch number: its parity picks the buffer
            const bool last = (jb == NB - 1) && (bt == NBAT - 1);
            const int jn = (bt + 1 < NBAT) ? jb : jb + 1, bn = (bt + 1 < NBAT) ? bt + 1 : 0;
            // lgkmcnt(0) BEFORE the next batch is issued: the current buffer is complete and the
            // new loads stay in flight during the FMAs below (0xc07f = vmcnt/expcnt untouched)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (!last) {
#pragma unroll
                for (int t = 0; t < 2 * JB; ++t) {
                    const double v = wv[(jn * FAN + 2 * bn) * JB + t];
                    if (g & 1) wa[t] = v; else wb[t] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * bt + u;
                if (i < FAN) {
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj)
                        if (jb * JB + jj < H)
                            acc[jj] = fma(in[i < FAN ? i : 0], (g & 1) ? wb[u * JB + jj] : wa[u * JB + jj], acc[jj]);
                }
            }
            if (bt == NBAT - 1) {
#pragma unroll
                for (int jj = 0; jj < JB; ++jj)
                    if (jb * JB + jj < H) out[jb * JB + jj] = tansig(acc[jj]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int K, int H, int NH>
__global__ __launch_bounds__(256, (H > 56 ? 1 : 2)) void score_valu_kernel(ScoreArgs A)
{
    constexpr int M = K * (K + 1) / 2;
    constexpr int DIN = K + M;
    constexpr int NB = (H + 7) / 8;
    const NetDev &net = A.net;
    const int64_t ntiles = (A.n + 255) / 256;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t c = tile * 256 + threadIdx.x;
        const bool valid = c < A.n;
        const int64_t cc = valid ? c : A.n - 1;
        Cand<K> cd;
        gather_candidate<K>(cd, A.set, A.n, cc, A.vars, A.Q, A.nv, A.L, (A.flags & SDPCUT_NN) != 0);
        const int32_t out_idx = A.orig[cc];
        double lam = 0.0;
        if (A.flags & SDPCUT_EIG) lam = candidate_eigmin<K>(cd);
        double obj = 0.0;
        if (A.flags & SDPCUT_NN) {
            cdouble_p inmap = (cdouble_p)net.inmap;
            double in[DIN];
#pragma unroll
            for (int i = 0; i < DIN; ++i) {
                const double v = (i < K) ? cd.x[i < K ? i : 0] : cd.q[i >= K ? i - K : 0];
                in[i] = (v - inmap[i]) * inmap[DIN + i] + net.ymin;
            }
            double a[H];
            dense_tansig<DIN, H>((cdouble_p)net.wvalu, (cdouble_p)net.bias, in, a);
            cdouble_p wv = (cdouble_p)net.wvalu + NB * DIN * 8;
#pragma unroll 1
            for (int l = 1; l < NH; ++l) {
                double o[H];
                dense_tansig<H, H>(wv, (cdouble_p)net.bias + l * 64, a, o);
#pragma unroll
                for (int j = 0; j < H; ++j) a[j] = o[j];
                wv += NB * H * 8;
            }
            cdouble_p wout = (cdouble_p)net.wout;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int j = 0; j + 1 < H; j += 2) {
                p0 = fma(a[j], wout[j], p0);
                p1 = fma(a[j + 1], wout[j + 1], p1);
            }
            if (H & 1) p0 = fma(a[H - 1], wout[H - 1], p0);
            {
#pragma clang fp contract(off)
                double acc = p0 + p1;
                acc = acc + net.b_out;
                const double y = (acc - net.y_ymin) / net.y_gain + net.y_xoffset;
                obj = cd.negSM;
                obj = obj + y * cd.max_elem;
            }
        }
        if (valid) {
            if (A.flags & SDPCUT_EIG) A.eig_out[out_idx] = lam;
            if (A.flags & SDPCUT_NN) A.obj_out[out_idx] = obj;
        }
        if (A.strong_out) {
            const unsigned long long m = __ballot(valid && obj > 0.0 && lam < SDPCUT_NEG_EIGVAL);
            if ((threadIdx.x & 63) == 0 && m)
                __hip_atomic_fetch_add((unsigned long long *)&A.strong_out[blockIdx.x & 7], (unsigned long long)__popcll(m),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Simple kernel: lane = candidate, activations in LDS, reference operation order.
template <int K>
__global__ __launch_bounds__(64) void score_simple_kernel(ScoreArgs A)
{
    constexpr int M = K * (K + 1) / 2;
    constexpr int DIN = K + M;
    __shared__ double act[2][MAX_HIDDEN][64];
    const int lane = threadIdx.x;
    const NetDev &net = A.net;
    const int64_t ntiles = (A.n + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t c = tile * 64 + lane;
        const bool valid = c < A.n;
        const int64_t cc = valid ? c : A.n - 1;
        Cand<K> cd;
        gather_candidate<K>(cd, A.set, A.n, cc, A.vars, A.Q, A.nv, A.L, (A.flags & SDPCUT_NN) != 0);
        const int32_t out_idx = A.orig[cc];
        double lam_c = 0.0;
        if (A.flags & SDPCUT_EIG) {
            const double lam = candidate_eigmin<K>(cd);
            lam_c = lam;
            if (valid) A.eig_out[out_idx] = lam;
        }
        if (!(A.flags & SDPCUT_NN)) continue;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int i = 0; i < DIN; ++i) {
                const double v = (i < K) ? cd.x[i < K ? i : 0] : cd.q[i >= K ? i - K : 0];
                act[0][i][lane] = (v - net.inmap[i]) * net.inmap[DIN + i] + net.ymin;
            }
            int cur = 0, fan_in = DIN;
            for (int l = 0; l < net.n_hidden; ++l) {
                const double *W = net.raw_w[l], *b = net.raw_b[l];
                for (int j = 0; j < net.width; ++j) {
                    double acc = 0.0;
                    for (int i = 0; i < fan_in; ++i) acc = acc + act[cur][i][lane] * W[j * fan_in + i];
                    acc = acc + b[j];
                    act[cur ^ 1][j][lane] = 2.0 / (libm_exp(acc * -2.0) + 1.0) + -1.0;      // the host libm's exp: NNs.so's bits (libm_exp.h)
                }
                cur ^= 1;
                fan_in = net.width;
            }
            const double *w = net.raw_w[net.n_hidden];
            double acc = 0.0;
            for (int j = 0; j < fan_in; ++j) acc = acc + act[cur][j][lane] * w[j];
            acc = acc + net.b_out;
            const double y = (acc - net.y_ymin) / net.y_gain + net.y_xoffset;
            double obj = cd.negSM;
            obj = obj + y * cd.max_elem;
            if (valid) A.obj_out[out_idx] = obj;
            if (A.strong_out) {
                const unsigned long long m = __ballot(valid && obj > 0.0 && lam_c < SDPCUT_NEG_EIGVAL);
                if (lane == 0 && m)
                    __hip_atomic_fetch_add((unsigned long long *)&A.strong_out[blockIdx.x & 7], (unsigned long long)__popcll(m),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Raw batched MLP forward on explicit inputs (the NNs.so call, batched) -- simple order.
__global__ __launch_bounds__(64) void nn_batch_kernel(NetDev net, int64_t count, const double *in, double *out)
{
#pragma clang fp contract(off)
    __shared__ double act[2][MAX_HIDDEN][64];
    const int lane = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t cc = c < count ? c : count - 1;
    const int DIN = net.d_in;
    for (int i = 0; i < DIN; ++i)
        act[0][i][lane] = (in[cc * DIN + i] - net.inmap[i]) * net.inmap[DIN + i] + net.ymin;
    int cur = 0, fan_in = DIN;
    for (int l = 0; l < net.n_hidden; ++l) {
        const double *W = net.raw_w[l], *b = net.raw_b[l];
        for (int j = 0; j < net.width; ++j) {
            double acc = 0.0;
            for (int i = 0; i < fan_in; ++i) acc = acc + act[cur][i][lane] * W[j * fan_in + i];
            acc = acc + b[j];
            act[cur ^ 1][j][lane] = 2.0 / (libm_exp(acc * -2.0) + 1.0) + -1.0;      // the host libm's exp: NNs.so's bits (libm_exp.h)
        }
        cur ^= 1;
        fan_in = net.width;
    }
    const double *w = net.raw_w[net.n_hidden];
    double acc = 0.0;
    for (int j = 0; j < fan_in; ++j) acc = acc + act[cur][j][lane] * w[j];
    acc = acc + net.b_out;
    if (c < count) out[c] = (acc - net.y_ymin) / net.y_gain + net.y_xoffset;
}

// ------------------------------------------------------------------------------------------
// Batched full eigen-decomposition of explicit sub-matrices (twin of _get_eigendecomp).
template <int K>
__global__ __launch_bounds__(64) void eig_batch_kernel(int64_t count, const double *xr, const double *Xr,
                                                       double *vals, double *vecs)
{
    constexpr int M = K * (K + 1) / 2;
    constexpr int D = K + 1;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    double x[K], X[M];
#pragma unroll
    for (int a = 0; a < K; ++a) x[a] = xr[i * K + a];
#pragma unroll
    for (int m = 0; m < M; ++m) X[m] = Xr[i * M + m];
    double a[D][D], v[D][D];
    fill_lifted<K>(a, x, X);
    jacobi_eig<D, true>(a, v);
    double w[D];
#pragma unroll
    for (int j = 0; j < D; ++j) w[j] = a[j][j];
    // ascending sort of the eigenpairs (odd-even transposition network, static indices)
#pragma unroll
    for (int round = 0; round < D; ++round)
#pragma unroll
        for (int j = round & 1; j + 1 < D; j += 2) {
            const bool sw = w[j + 1] < w[j];
            const double lo = sw ? w[j + 1] : w[j], hi = sw ? w[j] : w[j + 1];
            w[j] = lo; w[j + 1] = hi;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                const double p = v[r][j], qv = v[r][j + 1];
                v[r][j] = sw ? qv : p;
                v[r][j + 1] = sw ? p : qv;
            }
        }
#pragma unroll
    for (int j = 0; j < D; ++j) vals[i * D + j] = w[j];
    if (vecs) {
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int j = 0; j < D; ++j) vecs[(i * D + r) * D + j] = v[r][j];
    }
}

// ------------------------------------------------------------------------------------------
// Fragment-map probe: C[16][16] = A[16][4] * B[4][16] with the maps the MLP kernel assumes.
__global__ __launch_bounds__(64) void mfma_probe_kernel(const double *Am, const double *Bm, double *Cm)
{
    const int lane = threadIdx.x;
    const double a = Am[(lane & 15) * 4 + (lane >> 4)];   // A[row = lane&15][k = lane>>4]
    const double b = Bm[(lane >> 4) * 16 + (lane & 15)];  // B[k = lane>>4][col = lane&15]
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Cm[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

// ------------------------------------------------------------------------------------------
// host launchers
// workgroups per CU launched for the MFMA kernel (2 are resident at a time: 2 waves / SIMD)
#ifndef SDPCUT_MFMA_BLOCKS_PER_CU
#define SDPCUT_MFMA_BLOCKS_PER_CU 8
#endif

static int grid_for(sdpcut_ctx *h, int64_t ntiles, int per_cu)
{
    int64_t cap = (int64_t)h->n_cu * per_cu;
    int64_t g = ntiles < cap ? ntiles : cap;
    return (int)(g < 1 ? 1 : g);
}

// With SDPCUT_OPT_TIMING the kernel's own dispatch carries the two events (hipExtLaunchKernelGGL):
// its start / end timestamps are taken from the dispatch packet, without the two barrier packets
// and ~20 us per step that hipEventRecord around the launch costs.
#define SCORE_LAUNCH(kern, grid, block)                                                              \
    do {                                                                                             \
        if (ev_start || ev_stop)                                                                     \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, ev_start, ev_stop, 0, A);       \
        else                                                                                         \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, A);                             \
    } while (0)

// does size class K run on the MFMA / VALU kernels (network shape they are instantiated for)?
static bool net_shape_ok(const sdpcut_ctx *h, int K, uint32_t flags)
{
    if (!(flags & SDPCUT_NN)) return true;   // eig only: the network part of the kernel is skipped
    if (!h->net[K].set) return false;
    const NetDev &nd = h->net[K].dev;
    return (K == 2 && nd.width == 64 && nd.n_hidden == 3) || (K == 3 && nd.width == 50 && nd.n_hidden == 3) ||
           (K == 4 && nd.width == 50 && nd.n_hidden == 3) || (K == 5 && nd.width == 64 && nd.n_hidden == 4);
}

// The balanced tail round of ScoreArgs (see score_mfma_body): for lists whose last round-robin round would be at least 90 % full
// -- every resident slot then pays a whole strip for it -- the round is split evenly instead, three or four column tiles per wave,
// the four waves of a workgroup alike.  Emptier last rounds are left alone: the two waves of a SIMD share its issue slots, the
// dispatcher fills the gaps with the short workgroups, and a partial strip pays phase A for 64 lanes whatever it holds.  Measured
// (k = 3, this rule against none, profiles/r04_balanced_tail.txt): kernel time over the list length is a plateau up to a fill of
// 0.875 and steps up at 0.9; with the split the step becomes a ramp: 10^6 candidates (0.907) 302 -> 295 us, 0.95: 303 -> 300;
// forced at 0.75-0.875 the split LOSES 2-10 us, hence the threshold; the same at two and three full rounds.
#ifndef SDPCUT_BALANCED_TAIL
#define SDPCUT_BALANCED_TAIL 1
#endif
#ifndef SDPCUT_BALANCED_TAIL_MIN_FILL_PCT
#define SDPCUT_BALANCED_TAIL_MIN_FILL_PCT 90
#endif
static void set_balanced_tail(ScoreArgs &A, int grid)
{
    A.rr_end = A.n; A.tail_nhi = 0; A.tail_hi = 0; A.tail_lo = 0;
    if (!SDPCUT_BALANCED_TAIL || A.strip != 64) return;
    const int64_t W = (int64_t)grid * 4, round = W * 64;
    const int64_t R = A.n / round, rem = A.n - R * round;
    const int64_t tiles = (rem + 15) / 16;
    if (R < 1 || 100 * rem < SDPCUT_BALANCED_TAIL_MIN_FILL_PCT * round) return;
    const int64_t lo = tiles / W;      // 3 (a remainder of a whole round is R + 1 rounds)
    A.rr_end = R * round;
    A.tail_lo = (int32_t)lo;
    A.tail_hi = (int32_t)lo + 1;
    A.tail_nhi = (tiles - lo * W + 3) / 4 * 4;
}

// How far down its table a workgroup of `per_wg` candidates reports (TopkWs::pf_tab, topk_dev.h): to its m-th largest member,
// m = 24 + eight times its expected share E of a head of k out of n_total candidates.  A workgroup with MORE members of the head
// than that makes the selection fall back to its radix passes (the floor check).  For candidates in random order that never
// happens (a Poisson variable of mean 2.6 reaching 45); real covers are enumerated index set by index set, neighbours share
// variables and scores, and the head clusters: spar125-075-1 dim 4 has workgroups with > 18 of the 5000 (8 + 4 E failed there).
// Reporting more costs little: ~45 atomics per workgroup over a few hundred addresses.  0: no fine histogram (option off, or
// a table counter could overflow its 16 bits).
static int pf_mloc_for(const sdpcut_ctx *h, const ScoreFuse *fuse, int64_t per_wg, int K)
{
    if (!fuse || fuse->k <= 0 || per_wg >= 60000 || h->N < 1 || !pf_score_k(K)) return 0;
    const double share = (double)fuse->k * (double)per_wg / (double)h->N;
    const double m = 24.0 + 8.0 * share;
    return m > 60000.0 ? 60000 : (int)(m + 0.999);
}

template <int K>
static int launch_score_k(sdpcut_ctx *h, uint32_t flags, hipEvent_t ev_start, hipEvent_t ev_stop,
                          const ScoreFuse *fuse, int64_t *strong_out, hipStream_t st = nullptr)
{
    const Bucket &b = h->bucket[K];
    if (b.n == 0) return 0;
    if (!st) st = h->stream;
    ScoreArgs A;
    A.set = b.d_set; A.orig = b.d_orig; A.n = b.n;
    A.strip = 64;
    A.vars = h->d_vars; A.Q = h->d_Q; A.nv = h->nb_vars; A.L = h->L;
    A.eig_out = h->d_eig; A.obj_out = h->d_obj; A.flags = flags;
    A.tk = fuse ? (TopkWs *)fuse->ws : nullptr;
    A.tk_mode = fuse ? fuse->mode : 0;
    A.pf_mloc = 0;      // (set with the grid below)
    A.spread = 0;
    A.strong_out = ((flags & SDPCUT_EIG) && (flags & SDPCUT_NN)) ? strong_out : nullptr;
    A.net = h->net[K].dev;
    if ((flags & SDPCUT_NN) && !h->net[K].set)
        return sdpcut_fail(h, SDPCUT_ESTATE, "no network set for this candidate size");
    const bool mfma_ok = net_shape_ok(h, K, flags);
    if (h->kernel_variant == SDPCUT_KERNEL_VALU && mfma_ok) {
        const int64_t ntiles = (b.n + 255) / 256;
        const int grid = grid_for(h, ntiles, 8);
        if (K == 2) SCORE_LAUNCH((score_valu_kernel<2, 64, 3>), grid, 256);
        if (K == 3) SCORE_LAUNCH((score_valu_kernel<3, 50, 3>), grid, 256);
        if (K == 4) SCORE_LAUNCH((score_valu_kernel<4, 50, 3>), grid, 256);
        if (K == 5) SCORE_LAUNCH((score_valu_kernel<5, 64, 4>), grid, 256);
    } else if (h->kernel_variant == SDPCUT_KERNEL_MFMA && mfma_ok) {
        const int64_t ntiles = (b.n + 255) / 256;
        int grid = grid_for(h, ntiles, SDPCUT_MFMA_BLOCKS_PER_CU);
        if (b.n <= 32 * (int64_t)h->n_cu * 4 * 2) {      // part-empty device: one two-tile pass per wave
            A.strip = 32;
            grid = (int)(((b.n + 31) / 32 + 3) / 4);
        }
        set_balanced_tail(A, grid);
        A.pf_mloc = pf_mloc_for(h, fuse, (b.n + grid - 1) / grid, K);
        A.spread = A.pf_mloc > 0 && b.n <= (int64_t)grid * 4 * A.strip;
        if (A.tk && A.pf_mloc == 0) h->pf_counted = false;
        // (same arithmetic in every variant of one network: bit-equal scores)
#define SCORE_MFMA_LAUNCH(F, C)                                                             \
    do {                                                                                    \
        if (K == 2) SCORE_LAUNCH((score_mfma_kernel<2, 64, 3, F, C, mfma_cols(2)>), grid, 256);          \
        if (K == 3) SCORE_LAUNCH((score_mfma_kernel<3, 50, 3, F, C, mfma_cols(3)>), grid, 256);          \
        if (K == 4) SCORE_LAUNCH((score_mfma_kernel<4, 50, 3, F, C, mfma_cols(4)>), grid, 256);          \
        if (K == 5) SCORE_LAUNCH((score_mfma_kernel<5, 64, 4, F, C, mfma_cols(5)>), grid, 256);          \
    } while (0)
        const int f = A.tk ? A.tk_mode : 0;
        if (f != 0 && f != TK_MODE_FEAS && f != TK_MODE_OPT && f != TK_MODE_STRONG)
            return sdpcut_fail(h, SDPCUT_EINVAL, "score: no histogram variant for this selection mode");
        if (f == TK_MODE_FEAS ? !(flags & SDPCUT_EIG) : (f != 0 && !(flags & SDPCUT_NN)))
            return sdpcut_fail(h, SDPCUT_EINVAL, "score: the selection mode ranks by a measure this launch does not compute");
        if (A.net.unclamped_ok) {
            if (f == TK_MODE_STRONG) SCORE_MFMA_LAUNCH(TK_MODE_STRONG, false);
            else if (f == TK_MODE_OPT) SCORE_MFMA_LAUNCH(TK_MODE_OPT, false);
            else if (f == TK_MODE_FEAS) SCORE_MFMA_LAUNCH(TK_MODE_FEAS, false);
            else SCORE_MFMA_LAUNCH(0, false);
        } else {
            if (f == TK_MODE_STRONG) SCORE_MFMA_LAUNCH(TK_MODE_STRONG, true);
            else if (f == TK_MODE_OPT) SCORE_MFMA_LAUNCH(TK_MODE_OPT, true);
            else if (f == TK_MODE_FEAS) SCORE_MFMA_LAUNCH(TK_MODE_FEAS, true);
            else SCORE_MFMA_LAUNCH(0, true);
        }
#undef SCORE_MFMA_LAUNCH
    } else {
        const int64_t ntiles = (b.n + 63) / 64;
        const int grid = grid_for(h, ntiles, 16);
        SCORE_LAUNCH((score_simple_kernel<K>), grid, 64);
    }
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// One launch over all size classes of the list (score_mfma_all_kernel): possible when every class runs the MFMA kernels with
// the shipped kind of network (pre-activations provably bounded: the clamp-free instantiation, whose scores are what the
// per-class launches of such networks produce, bit for bit) and, with a fused selection, one of its histogram modes.
// -> 1 launched, 0 not applicable (the caller launches per class), < 0 error.
static int launch_classes_one(sdpcut_ctx *h, uint32_t flags, const ScoreFuse *fuse, int64_t *strong_out, hipEvent_t ev_start,
                              hipEvent_t ev_stop)
{
    if (h->kernel_variant != SDPCUT_KERNEL_MFMA || !(flags & SDPCUT_NN)) return 0;
    const int f = fuse ? fuse->mode : 0;
    if (f != 0 && f != TK_MODE_FEAS && f != TK_MODE_OPT && f != TK_MODE_STRONG) return 0;
    if (f == TK_MODE_FEAS && !(flags & SDPCUT_EIG)) return 0;
    ScoreArgsAll AA;
    AA.nclasses = 0;
    int order[SDPCUT_MAX_K - 1], m = 0;
    for (int k = 2; k <= SDPCUT_MAX_K; ++k)
        if (h->bucket[k].n > 0) {
            if (!h->net[k].set || !net_shape_ok(h, k, flags) || !h->net[k].dev.unclamped_ok) return 0;
            order[m++] = k;
        }
    for (int i = 1; i < m; ++i)      // the largest class first: its workgroups are the launch's critical path
        for (int j = i; j > 0 && h->bucket[order[j]].n > h->bucket[order[j - 1]].n; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int64_t blocks = 0;
    for (int i = 0; i < m; ++i) {
        const int k = order[i];
        const Bucket &b = h->bucket[k];
        ScoreArgs &A = AA.a[i];
        A.set = b.d_set; A.orig = b.d_orig; A.n = b.n;
        A.strip = 64;
        A.vars = h->d_vars; A.Q = h->d_Q; A.nv = h->nb_vars; A.L = h->L;
        A.eig_out = h->d_eig; A.obj_out = h->d_obj; A.flags = flags;
        A.tk = fuse ? (TopkWs *)fuse->ws : nullptr;
        A.tk_mode = f;
        A.pf_mloc = 0;
        A.spread = 0;
        A.strong_out = ((flags & SDPCUT_EIG) && (flags & SDPCUT_NN)) ? strong_out : nullptr;
        A.net = h->net[k].dev;
        int grid = grid_for(h, (b.n + 255) / 256, SDPCUT_MFMA_BLOCKS_PER_CU);
        if (b.n <= 32 * (int64_t)h->n_cu * 4 * 2) {      // (the same split as a launch over this class alone: launch_score_k)
            A.strip = 32;
            grid = (int)(((b.n + 31) / 32 + 3) / 4);
        }
        set_balanced_tail(A, grid);
        A.pf_mloc = pf_mloc_for(h, fuse, (b.n + grid - 1) / grid, k);
        A.spread = A.pf_mloc > 0 && b.n <= (int64_t)grid * 4 * A.strip;
        if (A.tk && A.pf_mloc == 0) h->pf_counted = false;
        blocks += grid;
        AA.k[i] = k;
        AA.bend[i] = (int32_t)blocks;
    }
    AA.nclasses = m;
    hipStream_t st = h->stream;
    const int grid = (int)blocks;
    ScoreArgsAll &A = AA;
    if (f == TK_MODE_STRONG) SCORE_LAUNCH((score_mfma_all_kernel<TK_MODE_STRONG, false>), grid, 256);
    else if (f == TK_MODE_OPT) SCORE_LAUNCH((score_mfma_all_kernel<TK_MODE_OPT, false>), grid, 256);
    else if (f == TK_MODE_FEAS) SCORE_LAUNCH((score_mfma_all_kernel<TK_MODE_FEAS, false>), grid, 256);
    else SCORE_LAUNCH((score_mfma_all_kernel<0, false>), grid, 256);
    HIP_TRY(h, hipGetLastError());
    return 1;
}

static int ensure_side_streams(sdpcut_ctx *h)
{
    if (h->ev_fork) return 0;
    // at the device's highest priority: the runtime keeps separate hardware queues per priority level, so the side streams do not
    // land in the queue of the handle's own (normal-priority) stream however many streams the process has created before -- in
    // a process that also runs torch they did, and the classes ran one after the other again, plus the events -- and their few
    // workgroups are dispatched ahead of the large class's waiting ones
    int least = 0, greatest = 0;
    HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int i = 0; i < 3; ++i) {
        HIP_TRY(h, hipStreamCreateWithPriority(&h->side_stream[i], hipStreamNonBlocking, greatest));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    return 0;
}

static int launch_score_any(sdpcut_ctx *h, int k, uint32_t flags, hipEvent_t es, hipEvent_t ee, const ScoreFuse *fuse, int64_t *strong_out,
                            hipStream_t st)
{
    return k == 2 ? launch_score_k<2>(h, flags, es, ee, fuse, strong_out, st)
         : k == 3 ? launch_score_k<3>(h, flags, es, ee, fuse, strong_out, st)
         : k == 4 ? launch_score_k<4>(h, flags, es, ee, fuse, strong_out, st)
                  : launch_score_k<5>(h, flags, es, ee, fuse, strong_out, st);
}

// the size classes one launch after the other on the handle's stream
static int launch_classes_seq(sdpcut_ctx *h, uint32_t flags, const ScoreFuse *fuse, int64_t *strong_out, const hipEvent_t *es,
                              const hipEvent_t *ee)
{
    for (int k = 2; k <= SDPCUT_MAX_K; ++k) {
        const int rc = launch_score_any(h, k, flags, es ? es[k] : nullptr, ee ? ee[k] : nullptr, fuse, strong_out, nullptr);
        if (rc) return rc;
    }
    return 0;
}

// Several size classes: real covers hold one large class and a few sets of the smaller sizes (spar100-050-1, dim 5: 72 673
// five-variable sets, 103 of four, 1 of three), and a launch over a hundred candidates costs what one pass costs -- 25-55 us of
// dependent stages -- however few they are.  Here the small classes go to side streams between a fork and a join event, the
// largest first on the handle's stream (its launch is what the round waits for).  Scores land in disjoint slots, the
// histograms are atomics: no order is needed.
static int launch_classes_side(sdpcut_ctx *h, uint32_t flags, const ScoreFuse *fuse, int64_t *strong_out, int kbig)
{
    int rc = ensure_side_streams(h);
    if (rc) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
    if ((rc = launch_score_any(h, kbig, flags, nullptr, nullptr, fuse, strong_out, nullptr))) return rc;
    int side = 0;
    for (int k = 2; k <= SDPCUT_MAX_K; ++k) {
        if (k == kbig || h->bucket[k].n == 0) continue;
        hipStream_t st = h->side_stream[side];
        HIP_TRY(h, hipStreamWaitEvent(st, h->ev_fork, 0));
        if ((rc = launch_score_any(h, k, flags, nullptr, nullptr, fuse, strong_out, st))) return rc;
        HIP_TRY(h, hipEventRecord(h->ev_join[side], st));
        ++side;
    }
    for (int i = 0; i < side; ++i) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_join[i], 0));
    return 0;
}

// Whether the side streams pay is a property of the process, not of the list: streams share a few hardware queues, assigned as
// they are created, and a side stream that lands in the queue of the handle's own stream runs its class BEHIND the large one --
// 193 instead of 154 us per round on spar070-050-1 in one process, 141 in another (profiles/r03_mixed_cover_round_timeline.txt).
// So the first multi-class scoring of a candidate list measures both forms (three plain scoring passes each, no histograms,
// the same scores written six times: ~1 ms once per list) and keeps the faster.
static int calibrate_side_streams(sdpcut_ctx *h, uint32_t flags, int kbig)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    float best[2] = {1e30f, 1e30f};
    int rc = 0;
    for (int rep = 0; rep < 3 && !rc; ++rep)
        for (int form = 0; form < 2 && !rc; ++form) {
            (void)hipEventRecord(e0, h->stream);
            rc = form ? launch_classes_side(h, flags, nullptr, nullptr, kbig) : launch_classes_seq(h, flags, nullptr, nullptr, nullptr, nullptr);
            (void)hipEventRecord(e1, h->stream);
            if (!rc && hipEventSynchronize(e1) != hipSuccess) rc = sdpcut_fail(h, SDPCUT_EHIP, "side-stream calibration failed");
            float ms = 0.f;
            if (!rc && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best[form]) best[form] = ms;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    h->side_choice = best[1] < 0.97f * best[0] ? 1 : 0;
    h->side_ms[0] = best[0];
    h->side_ms[1] = best[1];
    return 0;
}

int launch_score(sdpcut_ctx *h, uint32_t flags, const ScoreFuse *fuse, bool *fused, int64_t *strong_out)
{
    // (r5) did EVERY launch of this round count the fine histogram of the selection's class?  (The selection reads the table only then.)
    h->pf_counted = fuse != nullptr && fuse->k > 0;
    // the histograms are built by the MFMA kernel only: every non-empty size class must run on it
    if (fused) *fused = false;
    if (fuse) {
        bool ok = h->kernel_variant == SDPCUT_KERNEL_MFMA, any = false;
        for (int k = 2; k <= SDPCUT_MAX_K && ok; ++k) {
            if (h->bucket[k].n == 0) continue;
            ok = net_shape_ok(h, k, flags);
            any = true;
        }
        if (!ok || !any) fuse = nullptr;
        else if (fused) *fused = true;
    }
    // the first non-empty size class carries the start event, the last one the stop event
    int first = 0, last = 0;
    if (flags == SDPCUT_EIG && h->kernel_variant == SDPCUT_KERNEL_MFMA && h->eig_kernel) {
        // a pure-feasibility scan has its own kernel: one launch over all size classes (eig.hip)
        if (fuse && fuse->mode != TK_MODE_FEAS)
            return sdpcut_fail(h, SDPCUT_EINVAL, "score: the selection mode ranks by a measure this launch does not compute");
        h->timed_score = h->timing != 0 && h->N > 0;
        return launch_eig_only(h, fuse ? fuse->ws : nullptr, h->timed_score ? h->ev[0] : nullptr, h->timed_score ? h->ev[1] : nullptr,
                               fuse ? fuse->k : 0);
    }
    int nclasses = 0, kbig = 0;
    for (int k = 2; k <= SDPCUT_MAX_K; ++k)
        if (h->bucket[k].n > 0) {
            if (!first) first = k;
            last = k;
            ++nclasses;
            if (!kbig || h->bucket[k].n > h->bucket[kbig].n) kbig = k;
        }
    h->timed_score = h->timing && first;
    int rc;
    if (nclasses > 1 && h->one_launch) {
        rc = launch_classes_one(h, flags, fuse, strong_out, h->timed_score ? h->ev[0] : nullptr, h->timed_score ? h->ev[1] : nullptr);
        if (rc) return rc < 0 ? rc : 0;
    }
    if (nclasses > 1 && !h->timed_score && h->side_streams) {
        if (h->side_streams == 2 && h->side_choice < 0 && (rc = calibrate_side_streams(h, flags, kbig))) return rc;
        if (h->side_streams == 1 || h->side_choice == 1) return launch_classes_side(h, flags, fuse, strong_out, kbig);
    }
    hipEvent_t es[SDPCUT_MAX_K + 1] = {}, ee[SDPCUT_MAX_K + 1] = {};
    if (h->timed_score) { es[first] = h->ev[0]; ee[last] = h->ev[1]; }
    return launch_classes_seq(h, flags, fuse, strong_out, es, ee);
}

int launch_eig_batch(sdpcut_ctx *h, int k, int64_t count, const double *d_x, const double *d_X,
                     double *d_vals, double *d_vecs)
{
    if (count == 0) return 0;
    const int grid = (int)((count + 63) / 64);
    switch (k) {
    case 2: hipLaunchKernelGGL((eig_batch_kernel<2>), dim3(grid), dim3(64), 0, h->stream, count, d_x, d_X, d_vals, d_vecs); break;
    case 3: hipLaunchKernelGGL((eig_batch_kernel<3>), dim3(grid), dim3(64), 0, h->stream, count, d_x, d_X, d_vals, d_vecs); break;
    case 4: hipLaunchKernelGGL((eig_batch_kernel<4>), dim3(grid), dim3(64), 0, h->stream, count, d_x, d_X, d_vals, d_vecs); break;
    case 5: hipLaunchKernelGGL((eig_batch_kernel<5>), dim3(grid), dim3(64), 0, h->stream, count, d_x, d_X, d_vals, d_vecs); break;
    default: return sdpcut_fail(h, SDPCUT_EINVAL, "k must be 2..5");
    }
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int launch_nn_batch(sdpcut_ctx *h, int k, int64_t count, const double *d_in, double *d_out)
{
    if (count == 0) return 0;
    const int grid = (int)((count + 63) / 64);
    hipLaunchKernelGGL(nn_batch_kernel, dim3(grid), dim3(64), 0, h->stream, h->net[k].dev, count, d_in, d_out);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int launch_mfma_probe(sdpcut_ctx *h, const double *d_A, const double *d_B, double *d_C)
{
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, h->stream, d_A, d_B, d_C);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

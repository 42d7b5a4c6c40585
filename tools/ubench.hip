// Micro-benchmarks that size the design of the scoring kernel on gfx950:
//   fp64 MFMA vs fp64 VALU FMA rate, library tansig vs hand-rolled tansig, seed accuracy of
//   v_rcp_f64 / v_rsq_f64 / v_sqrt_f64.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)

__global__ __launch_bounds__(256) void mfma_rate(double *out, int iters)
{
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-6 + 1.0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_rate(double *out, int iters)
{
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3 + j;
    const double a = 1.0000001, b = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = fma(x[j], a, b);
    }
    double s = 0;
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ double tansig_lib(double n) { return 2.0 / (exp(-2.0 * n) + 1.0) - 1.0; }

// exp(y) for |y| <= 80: k = rint(y log2 e), r = y - k ln2 (two-term), degree-11 polynomial, ldexp
__device__ __forceinline__ double exp_fast(double y)
{
    const double k = rint(y * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, y);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 2.50521083854417187751e-08;              // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);          // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);          // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);          // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);          // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);          // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);          // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);          // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);          // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double tansig_fast(double n)      // = tansig() of csrc/score.hip
{
    const double y8 = fmin(n * -0.25, 88.0);
    const double k = rint(y8 * 11.541560327111707259);
    double r = fma(k, -8.66433975461404770613e-02, y8);
    r = fma(k, -2.38526866158823462503e-11, r);
    double p = 2.48015873015873015873e-05;
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    p = p * p;
    p = p * p;
    p = p * p;
    const double d = ldexp(p, (int)k) + 1.0;
    double q = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, q, 1.0);
    q = fma(q, fma(e, e, e), q);
    return fma(2.0, q, -1.0);
}

template <int V>
__global__ __launch_bounds__(256) void tansig_rate(double *out, int iters)
{
    double x[4];
    for (int j = 0; j < 4; ++j) x[j] = (threadIdx.x % 97) * 0.02 - 1.0 + j * 0.1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = (V == 0 ? tansig_lib(x[j]) : tansig_fast(x[j])) * 1.7 + 0.01 * j;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

__global__ void accuracy(const double *in, int n, double *o_lib, double *o_fast, double *o_rcp, double *o_rsq, double *o_sqrt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_lib[i] = tansig_lib(in[i]);
    o_fast[i] = tansig_fast(in[i]);
    double d = fabs(in[i]) + 0.5;
    o_rcp[i] = __builtin_amdgcn_rcp(d);
    o_rsq[i] = __builtin_amdgcn_rsq(d);
    o_sqrt[i] = __builtin_amdgcn_sqrt(d);
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    double *out;
    const int blocks = 256 * 8, threads = 256;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
    const int iters = 4000;
    for (int bl : {256 * 4, 256 * 8}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate, dim3(bl), dim3(threads), 0, 0, out, iters); }, 5);
        double flops = (double)bl * (threads / 64) * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4: blocks=%d  %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD at 2.4 GHz)\n", bl, ms,
               flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)bl * (threads / 64) * iters * 4 / 1024.0));
        ms = time_ms([&] { hipLaunchKernelGGL(fma_rate, dim3(bl), dim3(threads), 0, 0, out, iters); }, 5);
        flops = (double)bl * threads * iters * 8 * 2.0;
        printf("v_fma_f64       : blocks=%d  %.3f ms  %.2f TFLOP/s\n", bl, ms, flops / ms / 1e9);
    }
    {
        double ms0 = time_ms([&] { hipLaunchKernelGGL((tansig_rate<0>), dim3(blocks), dim3(threads), 0, 0, out, 500); }, 3);
        double ms1 = time_ms([&] { hipLaunchKernelGGL((tansig_rate<1>), dim3(blocks), dim3(threads), 0, 0, out, 500); }, 3);
        double evals = (double)blocks * threads * 500 * 4;
        printf("tansig library  : %.3f ms  %.1f G eval/s\n", ms0, evals / ms0 / 1e6);
        printf("tansig fast     : %.3f ms  %.1f G eval/s\n", ms1, evals / ms1 / 1e6);
    }
    {
        const int n = 1 << 20;
        std::vector<double> h(n), a(n), b(n), c(n), d(n), e(n);
        for (int i = 0; i < n; ++i) h[i] = (i & 1) ? -30.0 + 60.0 * i / n : -400.0 + 800.0 * i / n;   // fine + wide sweep
        double *din, *d0, *d1, *d2, *d3, *d4_;
        CHECK(hipMalloc(&din, n * 8)); CHECK(hipMalloc(&d0, n * 8)); CHECK(hipMalloc(&d1, n * 8));
        CHECK(hipMalloc(&d2, n * 8)); CHECK(hipMalloc(&d3, n * 8)); CHECK(hipMalloc(&d4_, n * 8));
        CHECK(hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(accuracy, dim3(n / 256), dim3(256), 0, 0, din, n, d0, d1, d2, d3, d4_);
        CHECK(hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(e.data(), d4_, n * 8, hipMemcpyDeviceToHost));
        double e_lib = 0, e_fast = 0, e_rcp = 0, e_rsq = 0, e_sqrt = 0;
        for (int i = 0; i < n; ++i) {
            const long double x = h[i];
            const long double t = 2.0L / (1.0L + expl(-2.0L * x)) - 1.0L;
            const double dd = std::fabs(h[i]) + 0.5;
            e_lib = std::fmax(e_lib, (double)fabsl(a[i] - t));
            e_fast = std::fmax(e_fast, (double)fabsl(b[i] - t));
            e_rcp = std::fmax(e_rcp, std::fabs(c[i] * dd - 1.0));
            e_rsq = std::fmax(e_rsq, std::fabs(d[i] * d[i] * dd - 1.0));
            e_sqrt = std::fmax(e_sqrt, std::fabs(e[i] * e[i] / dd - 1.0));
        }
        printf("max abs err tansig: library %.3e  fast %.3e\n", e_lib, e_fast);
        printf("seed rel err: v_rcp_f64 %.3e  v_rsq_f64 %.3e  v_sqrt_f64 %.3e\n", e_rcp, e_rsq, e_sqrt);
    }
    return 0;
}

// Does v_mfma_f64_16x16x4_f64 overlap with fp64 VALU work on gfx950, or do they share the fp64
// datapath?  (a) one wave interleaves 1 MFMA with K independent v_fma_f64; (b) MFMA-only waves
// and FMA-only waves share each SIMD.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int K, int NM>
__global__ __launch_bounds__(256) void mix(double *out, int iters)
{
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3 + j;
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-6 + 1.0;
    const double ca = 1.0000001, cb = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (NM) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) x[(m * K + j) & 15] = fma(x[(m * K + j) & 15], ca, cb);
        }
    }
    double s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512 threads: waves 0-3 MFMA only, waves 4-7 FMA only (one of each per SIMD)
template <int MODE>
__global__ __launch_bounds__(512) void split(double *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4) {
        if (MODE & 1) {
            d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-6 + 1.0;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
            for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        }
    } else {
        if (MODE & 2) {
            double x[16];
            for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3 + j;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int j = 0; j < 64; ++j) x[j & 15] = fma(x[j & 15], 1.0000001, 1e-9);
            for (int j = 0; j < 16; ++j) s += x[j];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main()
{
    double *out;
    if (hipMalloc(&out, 8 * 2048 * 512) != hipSuccess) return 1;
    const int it = 2000, bl = 1024;
#define RUN(K, NM) printf("1 wave/SIMD: %d MFMA + %2d FMA per step: %.3f ms\n", NM, K, \
    time_ms([&] { hipLaunchKernelGGL((mix<K, NM>), dim3(bl), dim3(256), 0, 0, out, it); }))
    RUN(0, 1); RUN(4, 1); RUN(8, 1); RUN(16, 1); RUN(24, 1); RUN(32, 1);
    RUN(4, 0); RUN(8, 0); RUN(16, 0); RUN(32, 0);
    printf("split waves: MFMA only %.3f ms | FMA only %.3f ms | both %.3f ms  (4 MFMA vs 64 FMA per step)\n",
           time_ms([&] { hipLaunchKernelGGL((split<1>), dim3(bl), dim3(512), 0, 0, out, it); }),
           time_ms([&] { hipLaunchKernelGGL((split<2>), dim3(bl), dim3(512), 0, 0, out, it); }),
           time_ms([&] { hipLaunchKernelGGL((split<3>), dim3(bl), dim3(512), 0, 0, out, it); }));
    return 0;
}

// Issue cost of the fp64 VALU instructions the scoring kernel is made of (gfx950), measured as
// wave-cycles per instruction with every SIMD saturated (8 independent registers per lane,
// 8 waves per SIMD).  Build on the GPU box: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X) \
    X(0, "v_fma_f64 (vgpr,vgpr,vgpr)", "v_fma_f64 %0, %0, %1, %2") \
    X(1, "v_fmac_f64 e32 (sgpr src0)", "v_fmac_f64_e32 %0, %3, %1") \
    X(2, "v_mul_f64", "v_mul_f64 %0, %0, %1") \
    X(3, "v_add_f64", "v_add_f64 %0, %0, %1") \
    X(4, "v_max_f64", "v_max_f64 %0, %0, %1") \
    X(5, "v_rcp_f64", "v_rcp_f64_e32 %0, %0") \
    X(6, "v_rsq_f64", "v_rsq_f64_e32 %0, %0") \
    X(7, "v_sqrt_f64", "v_sqrt_f64_e32 %0, %0") \
    X(8, "v_rndne_f64", "v_rndne_f64_e32 %0, %0") \
    X(9, "v_ldexp_f64", "v_ldexp_f64 %0, %0, %4") \
    X(10, "v_cvt_i32_f64 + v_cvt_f64_i32", "v_cvt_i32_f64_e32 %5, %0\n v_cvt_f64_i32_e32 %0, %5") \
    X(11, "v_fma_f32", "v_fma_f32 %5, %5, %5, %5") \
    X(12, "v_cndmask_b32 x2 (f64 select)", "v_cndmask_b32_e32 %5, %5, %5, vcc") \
    X(13, "v_exp_f32", "v_exp_f32_e32 %5, %5") \
    X(14, "v_rcp_f32", "v_rcp_f32_e32 %5, %5") \
    X(15, "v_mov_b64", "v_mov_b64_e32 %0, %1") \
    X(16, "v_cmp_lt_f64", "v_cmp_lt_f64_e32 vcc, %0, %1") \
    X(17, "v_div_scale_f64", "v_div_scale_f64 %0, vcc, %0, %1, %0") \
    X(18, "v_div_fmas_f64", "v_div_fmas_f64 %0, %0, %1, %2") \
    X(19, "v_div_fixup_f64", "v_div_fixup_f64 %0, %0, %1, %2") \
    X(20, "v_trig_preop/none: s_nop", "s_nop 0")

template <int OP>
__global__ __launch_bounds__(512) void op_rate(double *out, int iters, double sa)
{
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = 1.0 + threadIdx.x * 1e-3 + j * 0.01;
    const double b = 1.0000001, c = 1e-9;
    const int e = 0;
    float f = threadIdx.x * 1e-3f + 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define X(N, NAME, ASM) if (OP == N) asm volatile(ASM : "+v"(x[j]) : "v"(b), "v"(c), "s"(sa), "v"(e), "v"(f) : "vcc");
            OPS(X)
#undef X
        }
    }
    double s = f;
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, double *out)
{
    const int blocks = 1024, threads = 512, iters = 2000;   // 4 blocks/CU = 8 waves/SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((op_rate<OP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((op_rate<OP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    const double wave_instr_per_simd = (double)blocks * (threads / 64) * iters * 8 / 1024.0;
    printf("%-34s %8.3f ms   %6.2f ns/wave-instr/SIMD  = %5.2f cyc @2.4GHz\n", name, ms, ms * 1e6 / wave_instr_per_simd,
           ms * 1e6 / wave_instr_per_simd * 2.4);
}

int main()
{
    double *out;
    if (hipMalloc(&out, 8 * 1024 * 512) != hipSuccess) return 1;
#define X(N, NAME, ASM) run<N>(NAME, out);
    OPS(X)
#undef X
    return 0;
}

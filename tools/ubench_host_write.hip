// Can the host write the LP point straight into device memory (fine-grained allocation, large BAR) instead of staging it in pinned
// memory and launching a copy kernel?  And what does a kernel pay for gathering from such memory?   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void gather_kernel(const double *tab, int n, const int *idx, double *out, int m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) acc += tab[idx[i * 9 + j]];
    out[i] = acc;
}

int main()
{
    const int n = 5150, m = 1000000;
    double *fine = nullptr, *coarse = nullptr, *out = nullptr;
    int *idx = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&fine, n * sizeof(double), hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    CK(hipMalloc((void **)&coarse, n * sizeof(double)));
    CK(hipMalloc((void **)&out, m * sizeof(double)));
    CK(hipMalloc((void **)&idx, (size_t)m * 9 * sizeof(int)));
    std::vector<int> hidx((size_t)m * 9);
    unsigned s = 12345;
    for (auto &v : hidx) { s = s * 1664525u + 1013904223u; v = (int)((s >> 8) % n); }
    CK(hipMemcpy(idx, hidx.data(), hidx.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<double> host(n);
    for (int i = 0; i < n; ++i) host[i] = 0.001 * i;
    CK(hipMemcpy(coarse, host.data(), n * sizeof(double), hipMemcpyHostToDevice));
    hipPointerAttribute_t at;
    CK(hipPointerGetAttributes(&at, fine));
    printf("fine-grained pointer %p: type %d, hostPointer %p, devicePointer %p\n", (void *)fine, (int)at.type, at.hostPointer, at.devicePointer);
    // host write through the pointer itself
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double t_write = 0;
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < n; ++i) host[i] = 0.001 * i + rep;
        auto t0 = std::chrono::steady_clock::now();
        memcpy(fine, host.data(), n * sizeof(double));
        _mm_sfence();
        auto t1 = std::chrono::steady_clock::now();
        if (rep >= 100) t_write += std::chrono::duration<double, std::micro>(t1 - t0).count();
        hipLaunchKernelGGL(gather_kernel, dim3(8), dim3(256), 0, st, fine, n, idx, out, 2048);
        CK(hipStreamSynchronize(st));
        double chk[4];
        CK(hipMemcpy(chk, out, sizeof(chk), hipMemcpyDeviceToHost));
        double ref = 0;
        for (int j = 0; j < 9; ++j) ref += host[hidx[j]];
        if (chk[0] != ref) { printf("rep %d: kernel saw %.6f, host wrote %.6f -- STALE\n", rep, chk[0], ref); return 0; }
    }
    printf("host write of %d doubles into device memory + sfence: %.2f us; the kernel launched right behind it saw every value (200 rounds)\n", n, t_write / 100);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int which = 0; which < 2; ++which) {
        const double *tab = which ? fine : coarse;
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipEventRecord(a, st));
            hipLaunchKernelGGL(gather_kernel, dim3((m + 255) / 256), dim3(256), 0, st, tab, n, idx, out, m);
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("gather kernel (1e6 x 9 loads from a %d-double table), table in %s memory: %.1f us\n", n, which ? "FINE-grained" : "coarse-grained", best * 1e3);
    }
    return 0;
}

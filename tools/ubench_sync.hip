// Microbenchmark: how long after a kernel has finished does the host learn about it?
//   (a) hipStreamSynchronize, (b) polling a word the kernel's last instruction wrote into
//   device-mapped pinned host memory.  A ~300 us spin kernel stands in for the round.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_sync.hip -o /tmp/ubench_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>

__global__ void work_kernel(long long cycles, volatile uint64_t *flag, uint64_t seq, uint64_t *t_end)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (flag) {
            __threadfence_system();
            *flag = seq;
        }
        *t_end = wall_clock64();
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint64_t *h_flag, *d_flag, *d_tend;
    hipHostMalloc((void **)&h_flag, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void **)&d_flag, h_flag, 0);
    hipMalloc((void **)&d_tend, 8);
    *h_flag = 0;
    const long long cyc = 30000;      // wall_clock64 ticks at 100 MHz -> 300 us
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> tot;
        for (int it = 0; it < 300; ++it) {
            const uint64_t seq = (uint64_t)(mode * 1000 + it + 1);
            const double t0 = now_us();
            hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, cyc, mode ? d_flag : nullptr, seq, d_tend);
            if (mode == 0) {
                hipStreamSynchronize(st);
            } else if (mode == 1) {
                while (*(volatile uint64_t *)h_flag != seq) __builtin_ia32_pause();
            } else {
                while (*(volatile uint64_t *)h_flag != seq) __builtin_ia32_pause();
                hipStreamSynchronize(st);      // poll first, then the (now short) runtime sync
            }
            tot.push_back(now_us() - t0);
            if (mode == 1) hipStreamSynchronize(st);
        }
        std::sort(tot.begin(), tot.end());
        printf("%-28s launch->host-knows: median %.1f us  p10 %.1f  p90 %.1f  (kernel spins 300.0 us)\n",
               mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "poll mapped word" : "poll, then synchronize",
               tot[tot.size() / 2], tot[tot.size() / 10], tot[tot.size() * 9 / 10]);
    }
    return 0;
}

// Does a hipGraph shorten the device-side hand-off between DEPENDENT kernels that are already queued?
// A 300 us blocker kernel is launched first, then a chain of 6 small dependent kernels -- by stream launches
// (all enqueued while the blocker runs, as a round's launches are while the score kernel runs) or as one graph
// launch.  Timed with events around the chain (GPU side).  build: hipcc --offload-arch=gfx950 -O2 tools/ubench_graph.hip -o /tmp/ubench_graph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void blocker(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
__global__ void link(int *buf, int n)       // a few microseconds of dependent work
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = buf[(i * 7 + 1) % n] + 1;
}

int main()
{
    const int n = 1 << 16, chain = 6, reps = 200;
    int *buf; CK(hipMalloc(&buf, n * sizeof(int))); CK(hipMemset(buf, 0, n * sizeof(int)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // graph of the chain
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(link, dim3(n / 256), dim3(256), 0, s, buf, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int mode = 0; mode < 2; ++mode) {
        double sum = 0, best = 1e9;
        for (int r = 0; r < reps + 20; ++r) {
            hipLaunchKernelGGL(blocker, dim3(1), dim3(64), 0, s, 30000LL /* x10 ns = 300 us */, (int *)nullptr);
            CK(hipEventRecord(e0, s));
            if (mode == 0) for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(link, dim3(n / 256), dim3(256), 0, s, buf, n);
            else CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 20) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%s: chain of %d dependent kernels behind a busy queue: avg %.2f us, best %.2f us (%.2f us per link)\n",
               mode ? "graph launch " : "stream launch", chain, sum / reps * 1e3, best * 1e3, sum / reps * 1e3 / chain);
    }
    return 0;
}

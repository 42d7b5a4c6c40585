// Throughput of device-scope no-return atomics by address pattern (r5: what may the score kernels' fine histogram cost?).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_atomics tools/ubench_atomics.hip && /tmp/ubench_atomics
// Every workgroup (256 threads) issues `per_thread` atomic adds; thread t targets word pattern(t, i).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// mode 0: one address; 1: 32 words of one 128-B line; 2: 32 words in 32 different lines (stride 128 B);
// 3: 1024 contiguous words (4 KB, random-ish bin per op); 4: 1024 words padded to 64 B each; 5: 25 contiguous hot words;
// 6: 25 hot words padded to 128 B each; 7: 25 hot words, 8 replicas (by workgroup) contiguous
__global__ void k_atomics(unsigned *buf, int mode, int per_thread)
{
    unsigned s = blockIdx.x * 2654435761u + threadIdx.x * 40503u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = s >> 8;
        unsigned idx;
        switch (mode) {
        case 0: idx = 0; break;
        case 1: idx = r % 32; break;
        case 2: idx = (r % 32) * 32; break;
        case 3: idx = r % 1024; break;
        case 4: idx = (r % 1024) * 16; break;
        case 5: idx = r % 25; break;
        case 6: idx = (r % 25) * 32; break;
        default: idx = (blockIdx.x % 8) * 32 + r % 25; break;
        }
        __hip_atomic_fetch_add(&buf[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the retirement pattern: each workgroup does ~500 us of nothing?  no -- just the burst: grid workgroups x m ops by the first m lanes
__global__ void k_burst(unsigned *buf, int mode, int m)
{
    if ((int)threadIdx.x >= m) return;
    unsigned s = blockIdx.x * 2654435761u + threadIdx.x * 40503u + 777u;
    s = s * 1664525u + 1013904223u;
    const unsigned r = s >> 8;
    unsigned idx = mode == 5 ? r % 25 : mode == 6 ? (r % 25) * 32 : mode == 3 ? r % 1024 : (r % 1024) * 16;
    __hip_atomic_fetch_add(&buf[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main()
{
    unsigned *buf;
    hipMalloc(&buf, 1 << 20);
    hipMemset(buf, 0, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"one address", "32 words of one 128-B line", "32 words in 32 lines", "1024 contiguous words (4 KB)",
                           "1024 words padded to 64 B", "25 contiguous hot words", "25 hot words padded to 128 B", "25 hot words x 8 replicas"};
    for (int mode = 0; mode < 8; ++mode) {
        for (int grid : {64, 2048}) {
            const int per_thread = grid == 64 ? 64 : 4;
            k_atomics<<<grid, 256>>>(buf, mode, per_thread);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) k_atomics<<<grid, 256>>>(buf, mode, per_thread);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ops = 5.0 * grid * 256.0 * per_thread;
            printf("%-34s grid %4d x 256 x %2d: %8.1f us per launch, %7.2f ns per atomic (%.2f G/s)\n", names[mode], grid, per_thread,
                   ms * 1e3 / 5, ms * 1e6 / ops, ops / (ms * 1e6));
        }
    }
    for (int mode : {3, 4, 5, 6})
        for (int m : {4, 16, 64}) {
            k_burst<<<2048, 256>>>(buf, mode, m);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) k_burst<<<2048, 256>>>(buf, mode, m);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("burst: 2048 workgroups x %2d ops, %-30s %8.1f us per launch\n", m, names[mode], ms * 1e3 / 5);
        }
    return 0;
}

// Order-preserving images of fp64 scores used by the ranking kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// double -> u64, ascending; -0.0 is folded onto +0.0 first because Python compares them equal
__device__ __forceinline__ uint64_t key_of(double s)
{
    const uint64_t u = (uint64_t)__double_as_longlong(s + 0.0);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double score_of(uint64_t k)
{
    const uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// Counter-based candidate generator of the C4 workload (SURVEY.md section 8 d: "N = 1e8 triples
// generated on device from a counter-based RNG (Philox, seed 7, candidate id -> triple)").
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11), key = the 64-bit seed, counter = (candidate id low, id high, attempt, block).
// A candidate's index set is the sorted vector of k draws  v = floor(u32 * n / 2^32), accepted
// when all k are distinct -- the rejection step of synthetic.random_index_sets, which makes every
// k-subset equally likely -- else the attempt counter advances.  Plain functions usable from host
// and device code; tests/ hold a numpy twin (sdpcutsel_via_nn_amd/synthetic.py) of the same
// arithmetic.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define PHILOX_FN __host__ __device__ inline
#else
#define PHILOX_FN inline
#endif

#define PHILOX_MAX_ATTEMPTS 64

PHILOX_FN void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// index set of candidate `id` (k = 2..5 sorted distinct values in [0, n)); s has room for 5
PHILOX_FN void philox_index_set(uint64_t seed, uint64_t id, int n, int k, int32_t s[5])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (uint32_t attempt = 0; attempt < PHILOX_MAX_ATTEMPTS; ++attempt) {
        uint32_t r[8];
        philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), attempt, 0u, k0, k1, r);
        if (k > 4) philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), attempt, 1u, k0, k1, r + 4);
        for (int a = 0; a < 5; ++a) s[a] = a < k ? (int32_t)(((uint64_t)r[a] * (uint32_t)n) >> 32) : -1;
        // insertion sort of <= 5 values
        for (int a = 1; a < k; ++a) {
            const int32_t v = s[a];
            int b = a - 1;
            while (b >= 0 && s[b] > v) { s[b + 1] = s[b]; --b; }
            s[b + 1] = v;
        }
        bool distinct = true;
        for (int a = 1; a < k; ++a) distinct = distinct && s[a] != s[a - 1];
        if (distinct) return;
    }
    for (int a = 0; a < 5; ++a) s[a] = a < k ? a : -1;      // (probability < 1e-100 for n >= 2k)
}

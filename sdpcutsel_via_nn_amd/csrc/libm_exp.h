// exp() as the HOST's libm computes it -- for the batch-of-one / explicit-batch twins of the reference's NNs.so call only
// (nn_batch_kernel behind sdpcut_nn_batch and the six symbols of libsdpcut_nns.so), never for the scoring kernels.
//
// Provenance / licence: the operation sequence and the polynomial constants below are those of the GNU C Library's
// sysdeps/ieee754/dbl-64/e_exp.c (glibc 2.28+; derived from ARM Optimized Routines, Copyright (c) 2018 Arm Ltd and the FSF;
// glibc is licensed LGPL-2.1-or-later, the ARM original MIT / Apache-2.0 WITH LLVM-exception).  Nothing here comes from the
// reference repository.  The table is not copied: build.py recomputes 2^(i/128) in 60-digit decimals.
//
// What is claimed: bit-identity with a host whose libm is glibc >= 2.28 on an x86-64 CPU with FMA (the ifunc then
// picks the -mfma build, whose contractions are the fma() calls below) -- the build container and the GPU boxes; a host
// without FMA or with another libm runs NNs.so itself through other roundings, and tests/test_round4_cpu.py says so instead
// of failing there.  Arguments with |x| >= 512 leave glibc's main path (its `specialcase`: scaled evaluation near overflow /
// underflow) and take the device's exp() here; INSIDE tansig -- the only use, a = 2 / (1 + exp(-2n)) - 1 -- that cannot change a
// bit: exp(x) >= 2^738 gives 2 / (1 + e) < 2^-737, so a = -1 exactly, and exp(x) <= 2^-738 gives 1 + e = 1, a = 1 exactly,
// whatever the last bits of e are (inf and 0 included).
//
// NNs.so (MATLAB Coder) imports `exp` from libm; its outputs are therefore a function of glibc's algorithm:
// sysdeps/ieee754/dbl-64/e_exp.c (glibc >= 2.28, from ARM's optimized routines), EXP_TABLE_BITS = 7, polynomial of order 5, and
// on every x86-64 CPU with FMA the ifunc picks the variant compiled with -mfma, whose contractions are the fma() calls below.
// A fixed sequence of IEEE operations: the same bits on the device.  The 2^(i/128) table is recomputed at build time
// (build.py: libm_exp_table); the Python twin of this function reproduces the build container's libm bit for bit
// (tests/test_round4_cpu.py), and through it the 4 x 4096 NNs.so goldens are reproduced exactly on the GPU
// (tests/test_gpu_round4.py).
#pragma once
#include <hip/hip_runtime.h>

#include "_gen/libm_exp_table.inc"

__device__ __forceinline__ double libm_exp(double x)
{
#pragma clang fp contract(off)
    const unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    const unsigned abstop = (unsigned)(ix >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {                 // |x| < 2^-54 or |x| >= 512
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;   // tiny: exp(x) rounds like 1 + x
        return exp(x);      // (never reached by a tansig of the shipped networks: |n| <= 31.3, the argument is -2 n)
    }
    const double InvLn2N = 0x1.71547652b82fep0 * 128.0, Shift = 0x1.8p52;
    const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const double z = InvLn2N * x;
    double kd = z + Shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= Shift;
    const double r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x));
    const unsigned idx = 2u * (unsigned)(ki % 128ull);
    const unsigned long long top = ki << (52 - 7);
    const double tail = __longlong_as_double((long long)LIBM_EXP_TAB[idx]);
    const unsigned long long sbits = LIBM_EXP_TAB[idx + 1] + top;
    const double r2 = r * r;
    const double tmp = fma(r2 * r2, fma(r, C5, C4), fma(r2, fma(r, C3, C2), tail + r));
    const double scale = __longlong_as_double((long long)sbits);
    return fma(scale, tmp, scale);
}

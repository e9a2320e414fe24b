// Shared device-side types for libssamd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssamd {

// One pixel as the ASW kernels consume it: CIELab (float32) + the raw BGR bytes
// packed in one dword (byte3 = 0), 16 B so that one global_load_dwordx4 / one
// ds_read_b128 moves a pixel.  Replaces the reference's separate u8 image and
// fp64 Lab image (_passive.cpp:333-341).
struct __attribute__((aligned(16))) PixRec {
    float L, a, b;
    uint32_t bgrx;
};

// The last two operations of an ASW support weight (_passive.cpp:47-50, 71-74): proximity weight x exp(-colour distance / gammaC).
// SSAMD_W_FOLD=1 (experiment, tools/build_variants.sh): the proximity table holds log2 of the weights and the product becomes
// one fused multiply-add in the exponent -- one VALU instruction less per weight, the same result to within an ulp of the
// exponent.  Every ASW kernel goes through this one function, so they stay bit-identical to each other either way.
#ifndef SSAMD_W_FOLD
#define SSAMD_W_FOLD 0
#endif
__device__ __forceinline__ float asw_weight_finish(float dist, float kC, float prox)
{
#if SSAMD_W_FOLD
    return __builtin_amdgcn_exp2f(fmaf(dist, kC, prox));
#else
    return prox * __builtin_amdgcn_exp2f(dist * kC);
#endif
}

typedef unsigned long long u64;
static constexpr u64 KEY_NONE = ~0ull;

// WTA key: non-negative float cost bits in the high word (monotone as unsigned),
// candidate index in the low word, so that an unsigned 64-bit min implements
// "lowest cost, then lowest index" -- the reference's strict `<` scan keeps the
// first minimum, which is the smallest disparity in both passes
// (_passive.cpp:56,90-93 and 211,243-246).
__device__ __forceinline__ u64 make_key(float cost, uint32_t idx)
{
    return ((u64)__float_as_uint(cost) << 32) | (u64)idx;
}

}  // namespace ssamd

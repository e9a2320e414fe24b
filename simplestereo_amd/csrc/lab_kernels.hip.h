// K0: BGR u8 -> pixel records {L,a,b,bgrx}.  Pointwise, HBM-bound (3 B in, 16 B out
// per pixel).  Replaces ColorConversion::ImageFromBGR2Lab
// (reference headers/colorconversion.hpp:18-86, called at _passive.cpp:337-341).
#pragma once
#include "common.hip.h"
#include "glibc_math.hip.h"

namespace ssamd {

// sRGB byte -> linear*100 (colorconversion.hpp:19-37).  The transfer curve only
// depends on the byte, so it is a 256-entry table filled once by the host.
__constant__ float c_lin100[256];

// CIE f(t) with the reference's precision mix: float powf above the knee, double linear segment below
// (colorconversion.hpp:55-65).  Round 5: powf is glibc's own algorithm restated (glibc_math.hip.h: table-driven log2 / exp2 in
// double, one rounding to float) and returns the SAME float as the libm the reference links for every argument the conversion can
// produce -- checked exhaustively on the host (oracle/libm_check.c) -- so the records hold exactly the reference's Lab values
// rounded to float, and the fp64 tie-break pass the reference's doubles.  (Rounds 2-4 computed a correctly rounded cube root by
// Newton steps: 99.90 % of the floats were glibc's, whose powf is within 0.82 ulp but not correctly rounded.  -DSSAMD_LAB_NEWTON
// keeps that form for A/B builds.)
#ifdef SSAMD_LAB_NEWTON
__device__ __forceinline__ float lab_pow_third(float tf)
{
    const double t = (double)tf;
    const float l2 = __builtin_amdgcn_logf(tf);                          // v_log_f32: log2(t)
    float zf = __builtin_amdgcn_exp2f(l2 * -0.33333334f);              // t^(-1/3), ~1e-6
    zf = zf * fmaf(-tf, zf * zf * zf, 4.0f) * 0.33333334f;              // one step in fp32: ~1e-7
    double z = (double)zf;
    {
        const double z3 = z * z * z;
        z = z * fma(-t, z3, 4.0) * (1.0 / 3.0);                          // z <- z (4 - t z^3) / 3: ~1e-14
    }
    const double c = t * z * z;
    const double dy = (double)(float)(1 / 3.0) - 1.0 / 3.0;
    return (float)fma(c * dy, (double)(l2 * 0.6931471805599453f), c);
}
#else
__device__ __forceinline__ float lab_pow_third(float tf) { return glibc_powf_pos(tf, (float)(1 / 3.0)); }
#endif

// The conversion's three lookup tables (sRGB byte -> linear * 100; glibc powf's log2 and exp2 tables) staged in LDS by the
// record-producing kernels: 1.5 KB, indexed per lane.
struct LabTables {
    const float *lin100;
    const double *log2tab;
    const uint64_t *exp2tab;
};
#define SSAMD_LAB_TABLES_IN_LDS(name)                                                                          \
    __shared__ float name##_lin[256];                                                                          \
    __shared__ double name##_log2[32];                                                                         \
    __shared__ uint64_t name##_exp2[32];                                                                       \
    for (int k_ = threadIdx.x; k_ < 256; k_ += blockDim.x) name##_lin[k_] = c_lin100[k_];                      \
    for (int k_ = threadIdx.x; k_ < 32; k_ += blockDim.x) { name##_log2[k_] = gm_powf_log2_tab[k_]; name##_exp2[k_] = gm_exp2f_tab[k_]; } \
    __syncthreads();                                                                                           \
    const LabTables name{name##_lin, name##_log2, name##_exp2};

__device__ __forceinline__ double lab_f(double t, const LabTables &T)
{
#pragma clang fp contract(off)
#ifdef SSAMD_LAB_NEWTON
    if (t > 0.008856) return (double)lab_pow_third((float)t);
#else
    if (t > 0.008856) return (double)glibc_powf_pos_t((float)t, (float)(1 / 3.0), T.log2tab, T.exp2tab);
#endif
    return (7.787 * t) + (16.0 / 116.0);
}

// The reference's Lab values are DOUBLES (colorconversion.hpp:67-69: 116 * y - 16 ... on doubles that hold float powf results);
// the aggregation kernels use them rounded to float, the fp64 tie-break pass (asw_exact_kernels.hip.h) as they are.
__device__ __forceinline__ void bgr_to_lab_f64(uint32_t B, uint32_t G, uint32_t R, double &L, double &a, double &b, const LabTables &T)
{
#pragma clang fp contract(off)           // (the reference is a plain x86-64 build: no fused multiply-adds in its matrix and affine steps)
    const float r = T.lin100[R], g = T.lin100[G], bl = T.lin100[B];
    // observer 2 deg / D65 matrix in fp64 (colorconversion.hpp:40-42)
    const double X = r * 0.4124 + g * 0.3576 + bl * 0.1805;
    const double Y = r * 0.2126 + g * 0.7152 + bl * 0.0722;
    const double Z = r * 0.0193 + g * 0.1192 + bl * 0.9505;
    const float refX = 95.047f, refY = 100.0f, refZ = 108.883f;   // float constants, :48
    const double fx = lab_f(X / refX, T), fy = lab_f(Y / refY, T), fz = lab_f(Z / refZ, T);
    L = 116 * fy - 16;
    a = 500 * (fx - fy);
    b = 200 * (fy - fz);
}

__device__ __forceinline__ void bgr_to_lab(uint32_t B, uint32_t G, uint32_t R, float &L, float &a, float &b, const LabTables &T)
{
    double L64, a64, b64;
    bgr_to_lab_f64(B, G, R, L64, a64, b64, T);
    L = (float)L64;
    a = (float)a64;
    b = (float)b64;
}

// One thread per pixel; 4 consecutive pixels share 12 contiguous bytes but a 3-byte-per-lane read is still fully
// coalesced at the wave level (192 B/wave): one unaligned 4-byte read instead of three byte reads (the 4th byte is the
// next pixel's).
// Both images of a pair in ONE launch (round 3: a 1080p image is a 20 us kernel, two of them plus the gap between two
// dependent launches were a third of the pre-pass; at Tsukuba size the launch is most of it).
__global__ __launch_bounds__(256) void bgr2lab_records_pair_kernel(const uint8_t *__restrict__ bgrL, const uint8_t *__restrict__ bgrR,
                                                                   PixRec *__restrict__ recL, PixRec *__restrict__ recR, long long npix)
{
    SSAMD_LAB_TABLES_IN_LDS(T)
    long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    for (; q < 2 * npix; q += stride) {
        const bool right = q >= npix;
        const long long p = right ? q - npix : q;
        const uint8_t *const bgr = right ? bgrR : bgrL;
        uint32_t B, G, R;
        if (p + 1 < npix) {
            const uint32_t v = *reinterpret_cast<const u32_unaligned *>(bgr + 3 * p);
            B = v & 0xff; G = (v >> 8) & 0xff; R = (v >> 16) & 0xff;
        } else {
            B = bgr[3 * p]; G = bgr[3 * p + 1]; R = bgr[3 * p + 2];
        }
        PixRec o;
        bgr_to_lab(B, G, R, o.L, o.a, o.b, T);
        o.bgrx = B | (G << 8) | (R << 16);
        (right ? recR : recL)[p] = o;
    }
}

// debug/verification: plain float32 Lab image
__global__ __launch_bounds__(256) void bgr2lab_f32_kernel(const uint8_t *__restrict__ bgr,
                                                          float *__restrict__ lab, long long npix)
{
    SSAMD_LAB_TABLES_IN_LDS(T)
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; p < npix; p += stride) {
        float L, a, b;
        bgr_to_lab(bgr[3 * p], bgr[3 * p + 1], bgr[3 * p + 2], L, a, b, T);
        lab[3 * p] = L;
        lab[3 * p + 1] = a;
        lab[3 * p + 2] = b;
    }
}

}  // namespace ssamd

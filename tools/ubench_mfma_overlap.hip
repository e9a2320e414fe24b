// Does a bf16 / fp32 MFMA stream on some waves of a SIMD slow down fp32 VALU work on the other waves? (gfx950)
// Workgroup = 16 waves (4 per SIMD): waves 0-7 run N dependent-free v_fma_f32, waves 8-15 run M MFMAs (or nothing / VALU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_mfma_overlap ubench_mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { OTHER_IDLE, OTHER_VALU, OTHER_MFMA_BF16, OTHER_MFMA_F32 };
template <int OTHER, bool VALU_ON = true>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    if (wave < 8 && !VALU_ON) return;
    if (wave < 8 || OTHER == OTHER_VALU) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
        const float b = 1.0000001f, c = 1e-7f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
        if (s == 123.f) out[0] = s;
    } else if (OTHER == OTHER_MFMA_BF16) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + i); b[i] = (__bf16)(0.5f); }
        f32x16 acc0 = {}, acc1 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        }
        if (acc0[0] + acc1[0] == 123.f) out[1] = acc0[0];
    } else if (OTHER == OTHER_MFMA_F32) {
        float a = 1.5f, b = 0.5f;
        f32x16 acc0 = {}, acc1 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            }
        }
        if (acc0[0] + acc1[0] == 123.f) out[1] = acc0[0];
    }
}
template <int OTHER, bool VALU_ON = true>
float run(float *out, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OTHER, VALU_ON>), dim3(256), dim3(1024), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OTHER, VALU_ON>), dim3(256), dim3(1024), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float *out; (void)hipMalloc(&out, 16);
    const int iters = 20000;
    const float t0 = run<OTHER_IDLE>(out, iters), t1 = run<OTHER_VALU>(out, iters), t2 = run<OTHER_MFMA_BF16>(out, iters), t3 = run<OTHER_MFMA_F32>(out, iters);
    printf("2 VALU waves per SIMD, 64 v_fma_f32 x %d each:\n", iters);
    printf("  other 2 waves idle:                       %.3f ms\n", t0);
    printf("  other 2 waves the same VALU work:         %.3f ms\n", t1);
    printf("  other 2 waves 8 x mfma_f32_32x32x16_bf16: %.3f ms   (per iteration: 8 MFMAs = 8 x 32768 FLOP per wave)\n", t2);
    printf("  other 2 waves 8 x mfma_f32_32x32x2_f32:   %.3f ms\n", t3);
    printf("  MFMA waves alone (VALU waves exit): bf16 %.3f ms, f32 %.3f ms\n", run<OTHER_MFMA_BF16, false>(out, iters), run<OTHER_MFMA_F32, false>(out, iters));
    return 0;
}

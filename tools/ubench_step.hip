// tools/ubench_step.hip -- cost of one ASW tap column (32 taps) for variants of the inner loop,
// operands in registers (no LDS), 768-thread blocks = 3 waves/SIMD like the real kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(768, 3) void step_kernel(float* out, const float* in, int iters)
{
    v2f acc[4][8]; v2f ee[4][8]; float wl[4], wr[12];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) { acc[a][b] = v2f{0.f, 0.f}; ee[a][b] = v2f{in[(a * 8 + b) & 63], in[(a + b) & 63]}; }
    for (int k = 0; k < 4; ++k) wl[k] = in[k + threadIdx.x % 7];
    for (int k = 0; k < 12; ++k) wr[k] = in[k + threadIdx.x % 5];
    for (int it = 0; it < iters; ++it) {
        // keep the operands "fresh" so nothing is hoisted: opaque barriers on the weights
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(wl[k]));
        for (int k = 0; k < 12; ++k) asm volatile("" : "+v"(wr[k]));
        if (VAR == 0) {            // v_mul + v_pk_fma (current kernel)
#pragma unroll
            for (int di = 0; di < 8; ++di)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    const float w = wl[xi] * wr[7 + xi - di];
                    acc[xi][di] = __builtin_elementwise_fma(v2f{w, w}, ee[xi][di], acc[xi][di]);
                }
        } else if (VAR == 1) {     // 2 plain fma per tap
#pragma unroll
            for (int di = 0; di < 8; ++di)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    const float w = wl[xi] * wr[7 + xi - di];
                    acc[xi][di].x = fmaf(w, ee[xi][di].x, acc[xi][di].x);
                    acc[xi][di].y = fmaf(w, ee[xi][di].y, acc[xi][di].y);
                }
        } else if (VAR == 2) {     // only pk_fma (no product): lower bound of the accumulate part
#pragma unroll
            for (int di = 0; di < 8; ++di)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi)
                    acc[xi][di] = __builtin_elementwise_fma(v2f{wr[7 + xi - di], wr[7 + xi - di]}, ee[xi][di], acc[xi][di]);
        } else if (VAR == 3) {     // only the 32 multiplies
#pragma unroll
            for (int di = 0; di < 8; ++di)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) acc[xi][di].x += wl[xi] * wr[7 + xi - di];
        } else if (VAR == 4) {     // single fma per tap (cost only): what a (N)-only kernel would pay
#pragma unroll
            for (int di = 0; di < 8; ++di)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    const float w = wl[xi] * wr[7 + xi - di];
                    acc[xi][di].x = fmaf(w, ee[xi][di].x, acc[xi][di].x);
                }
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) s += acc[a][b].x + acc[a][b].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VAR> void run(float* out, const float* in, const char* name)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(step_kernel<VAR>, dim3(blocks), dim3(768), 0, 0, out, in, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(step_kernel<VAR>, dim3(blocks), dim3(768), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 3 waves x iters steps
    const double ns_per_step_per_simd = ms * 1e6 / (3.0 * iters);
    printf("%-44s %.3f ms  -> %.1f ns per 32-tap wave-step per SIMD (%.2f ns/tap)  => C3 (2.44e8 wave-steps/1024 SIMDs): %.1f ms\n",
           name, ms, ns_per_step_per_simd, ns_per_step_per_simd / 32, ns_per_step_per_simd * 2.44e8 / 1024 * 1e-6);
}

int main()
{
    float *out, *in; (void)hipMalloc(&out, 256 * 768 * 4); (void)hipMalloc(&in, 256 * 4);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.5f + i * 0.01f; (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(out, in, "v_mul + v_pk_fma(N,S')   [current]");
    run<1>(out, in, "v_mul + 2 x v_fma");
    run<2>(out, in, "v_pk_fma only");
    run<3>(out, in, "v_mul(+add) only");
    run<4>(out, in, "v_mul + 1 x v_fma (N only)");
    return 0;
}

// tools/ubench_step2.hip -- cost of one ASW tap column (32 taps) with packed fp32 forms that need no register moves
// (round 2).  768-thread blocks = 3 waves/SIMD like the real kernel; operands in registers.
//   VAR 0: v_mul_f32 + 2 x v_fma_f32 per tap                      (the kernel's form: 96 instructions per step)
//   VAR 1: v_mul_f32 + 1 x v_pk_fma_f32 per tap, w broadcast to both halves by op_sel (no v_mov)      (64)
//   VAR 2: 1 x v_pk_mul_f32 per 2 taps + 1 x v_pk_fma_f32 per tap, w picked from the product pair by op_sel (48)
//   VAR 3: as 2 with 20 instead of 16 v_pk_mul_f32 per step (the pairing the real index pattern allows)  (52)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_step2.hip -o tools/ubench_step2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(768, 3) void step_kernel(float* out, const float* in, int iters)
{
    v2f acc[8][4], ec[8][4];
    v2f wl[4], wr[6];                // 8 left weights, 12 right weights as aligned pairs
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) { acc[a][b] = v2f{0.f, 0.f}; ec[a][b] = v2f{in[(a * 4 + b) & 63], 40.f - in[(a * 4 + b) & 63]}; }
    for (int k = 0; k < 4; ++k) wl[k] = v2f{in[k + threadIdx.x % 7], in[k + 1 + threadIdx.x % 7]};
    for (int k = 0; k < 6; ++k) wr[k] = v2f{in[k + threadIdx.x % 5], in[k + 2 + threadIdx.x % 5]};
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(wl[k]));
        for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(wr[k]));
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) {
#pragma unroll
            for (int dp = 0; dp < 2; ++dp) {       // pairs of disparities
                const int k0 = (xi + 2 * dp) >> 1;  // some right-weight pair
                if (VAR == 0) {
                    const float wa = wl[xi >> 1][xi & 1] * wr[k0 % 6].x, wb = wl[xi >> 1][xi & 1] * wr[k0 % 6].y;
                    acc[xi][2 * dp].x = fmaf(wa, ec[xi][2 * dp].x, acc[xi][2 * dp].x);
                    acc[xi][2 * dp].y = fmaf(wa, ec[xi][2 * dp].y, acc[xi][2 * dp].y);
                    acc[xi][2 * dp + 1].x = fmaf(wb, ec[xi][2 * dp + 1].x, acc[xi][2 * dp + 1].x);
                    acc[xi][2 * dp + 1].y = fmaf(wb, ec[xi][2 * dp + 1].y, acc[xi][2 * dp + 1].y);
                } else if (VAR == 1) {
                    v2f wa, wb;          // only .x is written; the pair register is what v_pk_fma reads with op_sel lo/lo
                    wa.x = wl[xi >> 1][xi & 1] * wr[k0 % 6].x;
                    wb.x = wl[xi >> 1][xi & 1] * wr[k0 % 6].y;
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[xi][2 * dp]) : "v"(wa), "v"(ec[xi][2 * dp]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[xi][2 * dp + 1]) : "v"(wb), "v"(ec[xi][2 * dp + 1]));
                } else {
                    v2f wp;
                    if (xi & 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(wp) : "v"(wl[xi >> 1]), "v"(wr[k0 % 6]));
                    else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(wp) : "v"(wl[xi >> 1]), "v"(wr[k0 % 6]));
                    if (VAR == 3 && dp == 0 && (xi & 1)) {      // the 4 extra half-used products of the real pairing
                        v2f wq;
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(wq) : "v"(wl[xi >> 1]), "v"(wr[(k0 + 1) % 6]));
                        asm volatile("" :: "v"(wq));
                    }
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[xi][2 * dp]) : "v"(wp), "v"(ec[xi][2 * dp]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[xi][2 * dp + 1]) : "v"(wp), "v"(ec[xi][2 * dp + 1]));
                }
            }
        }
    }
    float s = 0;
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b].x + acc[a][b].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VAR> void run(float* out, const float* in, const char* name)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(step_kernel<VAR>, dim3(blocks), dim3(768), 0, 0, out, in, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(step_kernel<VAR>, dim3(blocks), dim3(768), 0, 0, out, in, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / (3.0 * iters);
    printf("%-52s %.3f ms -> %.1f ns per 32-tap wave-step per SIMD (%.2f ns/tap) => 1080p/193/35 taps alone: %.1f ms\n",
           name, best, ns, ns / 32, ns * 2.44e8 / 1024 * 1e-6);
}

int main()
{
    float *out, *in; (void)hipMalloc(&out, 256 * 768 * 4); (void)hipMalloc(&in, 256 * 4);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.5f + i * 0.01f; (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(out, in, "v_mul + 2 x v_fma                      [kernel]");
    run<1>(out, in, "v_mul + v_pk_fma (op_sel broadcast)");
    run<2>(out, in, "v_pk_mul per 2 taps + v_pk_fma (op_sel)");
    run<3>(out, in, "  same with 20 v_pk_mul per step");
    return 0;
}

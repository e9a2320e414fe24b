// GPU box: does the VGPR bank of an instruction's operands change the fp32 issue rate on gfx950?
//   hipcc --offload-arch=gfx950 -O2 -o ubench_vgpr_banks tools/ubench_vgpr_banks.hip && ./ubench_vgpr_banks
// 64 independent v_fma_f32 (one accumulator each, v8..v71) per loop trip in hand-allocated registers; the two
// multiplicands come from v0..v3 / v4..v7 (banks 0..3 = register index mod 4) in three placements relative to the
// accumulator's bank: all three operands in different banks, multiplicands sharing the accumulator's bank, and the two
// multiplicands sharing one bank.  Also v_mul_f32 + 2 x v_fma_f32 triples like the tap step (product in a rotating temp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void k(float *out, int trips)
{
    // the whole body is one asm block: registers v0..v75 are ours
    asm volatile(
        "v_mov_b32 v0, 1.0\n v_mov_b32 v1, 0.5\n v_mov_b32 v2, 2.0\n v_mov_b32 v3, 4.0\n"
        "v_mov_b32 v4, 0.5\n v_mov_b32 v5, 1.0\n v_mov_b32 v6, 0.5\n v_mov_b32 v7, 2.0\n"
        ".set i, 8\n .rept 68\n v_mov_b32 v[i], 0\n .set i, i+1\n .endr\n"
        "s_mov_b32 s20, %1\n"
        "1:\n"
        ".set i, 0\n"
        ".rept 64\n"
        ".if %2 == 0\n"       // all different banks
        "  v_fma_f32 v[8+i], v[(i+1)&3], v[4+((i+2)&3)], v[8+i]\n"
        ".elseif %2 == 1\n"   // everything in the accumulator's bank
        "  v_fma_f32 v[8+i], v[i&3], v[4+(i&3)], v[8+i]\n"
        ".elseif %2 == 2\n"   // multiplicands share a bank, accumulator elsewhere
        "  v_fma_f32 v[8+i], v[(i+1)&3], v[4+((i+1)&3)], v[8+i]\n"
        ".elseif %2 == 3\n"   // v_mul distinct banks
        "  v_mul_f32 v[8+i], v[(i+1)&3], v[4+((i+2)&3)]\n"
        ".elseif %2 == 4\n"   // v_mul same bank
        "  v_mul_f32 v[8+i], v[(i+1)&3], v[4+((i+1)&3)]\n"
        ".elseif %2 == 5\n"   // tap-like triple, banks spread: w = a*b ; N += w*e ; S += w*c   (i counts triples: 21 per trip + 1)
        "  .if i < 21\n"
        "  v_mul_f32 v[72+(i&3)], v[(i+1)&3], v[4+((i+2)&3)]\n"
        "  v_fmac_f32 v[8+2*i], v[72+(i&3)], v[4+((i+3)&3)]\n"
        "  v_fmac_f32 v[9+2*i], v[72+(i&3)], v[(i+2)&3]\n"
        "  .endif\n"
        ".endif\n"
        ".set i, i+1\n"
        ".endr\n"
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        "v_add_f32 v8, v8, v9\n"
        "global_store_dword %0, v8, off\n"
        "s_waitcnt vmcnt(0)\n"
        :
        : "v"(out + blockIdx.x * blockDim.x + threadIdx.x), "s"(trips), "n"(VARIANT)
        : "memory", "scc", "s20",
          "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19",
          "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
          "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59",
          "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75");
}

int main()
{
    float *d;
    CHECK(hipMalloc(&d, 1 << 26));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int trips = 20000;
    const char *names[] = {"v_fma_f32, three operands in three banks", "v_fma_f32, all operands in the accumulator's bank",
                           "v_fma_f32, multiplicands share a bank", "v_mul_f32, operands in two banks", "v_mul_f32, operands in one bank",
                           "v_mul + 2 v_fmac (tap-like, banks spread)"};
    void (*ks[])(float *, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>};
    const int per_trip[] = {64, 64, 64, 64, 64, 63};
    for (int v = 0; v < 6; ++v)
        for (int wps : {1, 2, 3, 4}) {       // waves per SIMD: 256 CUs x 4 SIMDs x wps waves, as 256-thread blocks
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(ks[v], dim3(blocks), dim3(256), 0, 0, d, 100);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(ks[v], dim3(blocks), dim3(256), 0, 0, d, trips);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double instr_per_simd = (double)trips * per_trip[v] * wps;
            printf("%-52s waves/SIMD=%d  %.3f ms  %.3f G wave-instr/s/SIMD  (%.2f cycles per instruction at 2.4 GHz)\n", names[v], wps, ms,
                   instr_per_simd / (ms * 1e-3) / 1e9, 2.4e9 * ms * 1e-3 / instr_per_simd);
        }
    return 0;
}

// tools/ubench_valu.hip -- instruction issue-rate microbenchmark for gfx950.
// Measures, per SIMD, the cycles one wave64 instruction of each kind occupies when
// 1/2/4 waves per SIMD issue long runs of independent instructions.  Used to price
// the ASW inner loop (DESIGN.md "VALU budget").  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { FMA, MUL, ADD, PKFMA, PKMUL, PKADD, CVTUB, FMAMIX, DOT2, PKFMAH, SAD, EXP, SQRT, MIX_PKMUL_FMAMIX_PKADD, DSR128, DSR64, DSR32, NKIND };
static const char* names[NKIND] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
  "v_cvt_f32_ubyte1", "v_fma_mix_f32", "v_dot2_f32_f16", "v_pk_fma_f16", "v_sad_u8", "v_exp_f32", "v_sqrt_f32",
  "mix(1 pk_mul+2 fma_mix+1 pk_add)", "ds_read_b128", "ds_read_b64", "ds_read_b32"};

template <int K>
__global__ __launch_bounds__(256) void ub(float* out, long long* cyc, int iters)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    float a[16]; f2 p[16]; h2 hh[16]; unsigned u[16];
    const float b = threadIdx.x * 1e-9f + 1.0f, c = 1e-7f;
    const f2 b2 = {b, b}, c2 = {c, c};
    const h2 hb = {(_Float16)1.0f, (_Float16)0.5f};
    for (int k = 0; k < 16; ++k) { a[k] = k + threadIdx.x; p[k] = f2{a[k], a[k] + 1}; hh[k] = h2{(_Float16)k, (_Float16)1}; u[k] = threadIdx.x * 2654435761u + k; }
    const unsigned laddr = (threadIdx.x * 16) & 16383;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define I_FMA(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MUL(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define I_ADD(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(c));
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(b2), "v"(c2));
#define I_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(b2));
#define I_PKADD(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(c2));
#define I_CVT(k) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[k]) : "v"(u[k]));
#define I_FMAMIX(k) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(a[k]) : "v"(b), "v"(hh[k]));
#define I_DOT2(k) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[k]) : "v"(hh[k]), "v"(hb));
#define I_PKFMAH(k) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(hh[k]) : "v"(hb), "v"(hb));
#define I_SAD(k) asm volatile("v_sad_u8 %0, %1, %2, 0" : "=v"(u[k]) : "v"(u[k]), "v"(u[(k + 1) & 15]));
#define I_EXP(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
#define I_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define I_MIXED(k) asm volatile("v_pk_mul_f32 %0, %3, %0\n v_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_pk_add_f32 %0, %6, %0" \
        : "+v"(p[k]), "+v"(a[k]), "+v"(a[(k + 8) & 15]) : "v"(b2), "v"(b), "v"(hh[k]), "v"(c2));
        if (K == FMA) { REP16(I_FMA) REP16(I_FMA) }
        if (K == MUL) { REP16(I_MUL) REP16(I_MUL) }
        if (K == ADD) { REP16(I_ADD) REP16(I_ADD) }
        if (K == PKFMA) { REP16(I_PKFMA) REP16(I_PKFMA) }
        if (K == PKMUL) { REP16(I_PKMUL) REP16(I_PKMUL) }
        if (K == PKADD) { REP16(I_PKADD) REP16(I_PKADD) }
        if (K == CVTUB) { REP16(I_CVT) REP16(I_CVT) }
        if (K == FMAMIX) { REP16(I_FMAMIX) REP16(I_FMAMIX) }
        if (K == DOT2) { REP16(I_DOT2) REP16(I_DOT2) }
        if (K == PKFMAH) { REP16(I_PKFMAH) REP16(I_PKFMAH) }
        if (K == SAD) { REP16(I_SAD) REP16(I_SAD) }
        if (K == EXP) { REP16(I_EXP) REP16(I_EXP) }
        if (K == SQRT) { REP16(I_SQRT) REP16(I_SQRT) }
        if (K == MIX_PKMUL_FMAMIX_PKADD) { I_MIXED(0) I_MIXED(1) I_MIXED(2) I_MIXED(3) I_MIXED(4) I_MIXED(5) I_MIXED(6) I_MIXED(7) }
        if (K == DSR128) {
#define I_D128(k) { f4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(laddr), "n"(k * 16)); asm volatile("" :: "v"(v)); }
            REP16(I_D128) REP16(I_D128) asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (K == DSR64) {
#define I_D64(k) { f2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(laddr / 2), "n"(k * 8)); asm volatile("" :: "v"(v)); }
            REP16(I_D64) REP16(I_D64) asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (K == DSR32) {
#define I_D32(k) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(laddr / 4), "n"(k * 4)); asm volatile("" :: "v"(v)); }
            REP16(I_D32) REP16(I_D32) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    long long t1 = clock64();
    float s = 0; for (int k = 0; k < 16; ++k) s += a[k] + p[k].x + p[k].y + (float)hh[k].x + u[k];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K> void run(float* out, long long* cyc, int iters)
{
    for (int wps : {1, 2, 4}) {            // waves per SIMD: blocks of 256 threads = 1 wave/SIMD each
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(ub<K>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(ub<K>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= blocks;
        const int per_iter = (K == MIX_PKMUL_FMAMIX_PKADD) ? 32 : 32;
        const double instr_per_wave = (double)iters * per_iter;
        // wall-clock based: total wave-instr per SIMD / time
        const double simds = 1024.0;
        const double wave_instr_per_simd = instr_per_wave * wps * (blocks * 4.0 / (simds * wps));
        printf("%-36s waves/SIMD=%d  clock64/instr/wave=%.2f  -> per-SIMD issue interval=%.2f ticks   wall: %.3f ms, %.2f G wave-instr/s/SIMD\n",
               names[K], wps, mean / instr_per_wave, mean / instr_per_wave / wps, ms, wave_instr_per_simd / (ms * 1e-3) / 1e9);
    }
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    const int iters = 20000;
    run<FMA>(out, cyc, iters); run<MUL>(out, cyc, iters); run<ADD>(out, cyc, iters);
    run<PKFMA>(out, cyc, iters); run<PKMUL>(out, cyc, iters); run<PKADD>(out, cyc, iters);
    run<CVTUB>(out, cyc, iters); run<FMAMIX>(out, cyc, iters); run<DOT2>(out, cyc, iters); run<PKFMAH>(out, cyc, iters);
    run<SAD>(out, cyc, iters); run<EXP>(out, cyc, iters); run<SQRT>(out, cyc, iters);
    run<MIX_PKMUL_FMAMIX_PKADD>(out, cyc, iters);
    run<DSR128>(out, cyc, iters / 4); run<DSR64>(out, cyc, iters / 4); run<DSR32>(out, cyc, iters / 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("device: %s  CUs=%d  clockRate=%d kHz  wallClockRate=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate, pr.clockInstructionRate);
    return 0;
}

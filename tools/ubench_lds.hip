// LDS read-throughput microbenchmark (gfx950): ns per wave-wide LDS read per CU for several widths and address
// patterns; 12 waves per CU, each issuing 8 independent reads (inline asm, so the compiler cannot merge or narrow
// them) before one s_waitcnt.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_lds ubench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
enum { B128_CONSEC, B128_BCAST, B128_WL5, B128_WR5, B128_WL49, B128_WR49, B128_STRIDE32, B96_PACKED, B96_STRIDE16, B64_CONSEC, B64_STRIDE16, B32_CONSEC, B32_STRIDE16, NPAT };
static const char *names[NPAT] = {"b128 lane*16", "b128 broadcast", "b128 wl (lane/5)*32", "b128 wr 32xg-16dg (DG 5)", "b128 wl (lane/49)*32",
                                  "b128 wr 32xg-16dg (DG 49)", "b128 lane*32", "b96 lane*12", "b96 lane*16", "b64 lane*8", "b64 lane*16", "b32 lane*4", "b32 lane*16"};
#define RD(INSTR, T, N)                                                                                       \
    {                                                                                                         \
        T v0, v1, v2, v3, v4, v5, v6, v7;                                                                     \
        asm volatile(INSTR " %0, %8\n" INSTR " %1, %8 offset:1024\n" INSTR " %2, %8 offset:2048\n" INSTR " %3, %8 offset:3072\n" \
                     INSTR " %4, %8 offset:4096\n" INSTR " %5, %8 offset:5120\n" INSTR " %6, %8 offset:6144\n" INSTR " %7, %8 offset:7168\n" \
                     "s_waitcnt lgkmcnt(0)"                                                                   \
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory"); \
        sink(v0); sink(v1); sink(v2); sink(v3); sink(v4); sink(v5); sink(v6); sink(v7);                       \
    }
typedef float f3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void sink(float v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void sink(float2 v) { asm volatile("" ::"v"(v.x), "v"(v.y)); }
__device__ __forceinline__ void sink(f3 v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z)); }
__device__ __forceinline__ void sink(float4 v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
template <int PAT>
__global__ __launch_bounds__(256, 3) void k(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 10240; i += 256) reinterpret_cast<float *>(smem)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int off = lane * 16;
    if (PAT == B128_BCAST) off = 0;
    if (PAT == B128_WL5) off = (lane / 5) * 32;
    if (PAT == B128_WR5) off = 32 * (lane / 5) - 16 * (lane % 5) + 64;
    if (PAT == B128_WL49) off = (lane / 49) * 32;
    if (PAT == B128_WR49) off = 32 * (lane / 49) - 16 * (lane % 49) + 784;
    if (PAT == B128_STRIDE32) off = lane * 32 % 1024 + (lane / 32) * 16;
    if (PAT == B96_PACKED) off = lane * 12;
    if (PAT == B64_CONSEC) off = lane * 8;
    if (PAT == B32_CONSEC) off = lane * 4;
    const unsigned addr = (unsigned)(size_t)(smem + wave * 8192 + off) & 0xffffu;      // LDS byte address
    for (int it = 0; it < iters; ++it) {
        if constexpr (PAT <= B128_STRIDE32) RD("ds_read_b128", float4, 0)
        else if constexpr (PAT <= B96_STRIDE16) RD("ds_read_b96", f3, 0)
        else if constexpr (PAT <= B64_STRIDE16) RD("ds_read_b64", float2, 0)
        else RD("ds_read_b32", float, 0)
    }
    if (iters < 0) out[0] = 1.f;
}
template <int PAT>
void run(float *out)
{
    const int iters = 20000, wgs = 256 * 3;     // 3 workgroups of 4 waves per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 40960, 0, out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 40960, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double reads_per_cu = 12.0 * iters * 8;
    printf("%-28s %.3f ms, %.2f ns per wave-wide read per CU\n", names[PAT], ms, ms * 1e6 / reads_per_cu);
    if constexpr (PAT + 1 < NPAT) run<PAT + 1>(out);
}
int main() { float *out; (void)hipMalloc(&out, 4); run<0>(out); return 0; }

// tools/ubench_unpack.hip -- what the e-dword unpack of an ASW tap step costs, and whether it can be had cheaper (round 4).
// One step = 32 taps x {v_mul_f32, 2 x v_fma_f32} + the unpack of ONE packed dword (4 bytes e = min(40, |dB|+|dG|+|dR|)) into the
// (e, 40 - e) operands of the thread's register window.  768-thread blocks = 3 waves per SIMD like the phase-shifted kernel.
//   VAR 0: no unpack (operands stay in registers)                                    -- the floor (tools/ubench_step2.hip VAR 0)
//   VAR 1: 4 x v_cvt_f32_ubyteN + 4 x v_sub_f32                                       -- the kernels' form until round 3
//   VAR 2: 4 x {v_and / v_bfe_u32 / v_lshrrev} + 4 x v_sub_u32: the bytes stay INTEGERS, i.e. the fp32 denormals e * 2^-149, and the
//          support weights carry 2^63 each (folded into the proximity table), so w' * e' = w * e * 2^-23 with the same rounding
//   VAR 3: 1 x v_sub_u32 on the packed dword (0x28282828 - e4) + 8 x byte extraction
// Also checks that the denormal form is EXACT: fma(wl * 2^63 * wr * 2^63, e as denormal bits, acc * 2^-23) == fma(wl * wr, e, acc) * 2^-23
// bit for bit (fp32 denormal inputs are honoured by v_fma_f32 / v_mul_f32: the kernels are built with denorm mode 3).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_unpack.hip -o tools/ubench_unpack
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

template <int VAR>
__global__ __launch_bounds__(768, 3) void step_kernel(float *out, const float *in, const uint32_t *ein, int iters)
{
    float accN[8][4], accS[8][4], e[8][4], c[8][4], wl[8], wr[12];
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 4; ++b) { accN[a][b] = 0.f; accS[a][b] = 0.f; e[a][b] = in[(a * 4 + b) & 63]; c[a][b] = 40.f - e[a][b]; }
    for (int k = 0; k < 8; ++k) wl[k] = in[k + threadIdx.x % 7];
    for (int k = 0; k < 12; ++k) wr[k] = in[k + threadIdx.x % 5];
    uint32_t pk = ein[threadIdx.x & 63];
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(wl[k]));
        for (int k = 0; k < 12; ++k) asm volatile("" : "+v"(wr[k]));
        asm volatile("" : "+v"(pk));
        const int slot = it & 7;     // (the real kernel rotates the window slot at compile time; one fixed slot prices the same work)
        (void)slot;
        if (VAR == 1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) { const float v = (float)((pk >> (8 * b)) & 0xffu); e[0][b] = v; c[0][b] = 40.f - v; }
        } else if (VAR == 2) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t v = (pk >> (8 * b)) & 0xffu;
                e[0][b] = __uint_as_float(v); c[0][b] = __uint_as_float(40u - v);
            }
        } else if (VAR == 3) {
            const uint32_t ck = 0x28282828u - pk;
#pragma unroll
            for (int b = 0; b < 4; ++b) { e[0][b] = __uint_as_float((pk >> (8 * b)) & 0xffu); c[0][b] = __uint_as_float((ck >> (8 * b)) & 0xffu); }
        }
#pragma unroll
        for (int xi = 0; xi < 8; ++xi)
#pragma unroll
            for (int di = 0; di < 4; ++di) {
                const float w = wl[xi] * wr[xi - di + 3];
                accN[xi][di] = fmaf(w, e[xi][di], accN[xi][di]);
                accS[xi][di] = fmaf(w, c[xi][di], accS[xi][di]);
            }
    }
    float s = 0;
    for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) s += accN[a][b] + accS[a][b];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// exactness of the denormal form over random weights and all 41 byte values
__global__ void exact_kernel(const float *wl, const float *wr, const float *acc, uint32_t *bad, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float S = 9223372036854775808.f;               // 2^63
    const float D = 1.1920928955078125e-07f;              // 2^-23
    for (uint32_t ev = 0; ev <= 40; ++ev) {
        const float w = wl[i] * wr[i];
        const float n0 = fmaf(w, (float)ev, acc[i]), s0 = fmaf(w, 40.f - (float)ev, acc[i]);
        const float ws = (wl[i] * S) * (wr[i] * S);
        const float n1 = fmaf(ws, __uint_as_float(ev), acc[i] * D), s1 = fmaf(ws, __uint_as_float(40u - ev), acc[i] * D);
        if (n1 != n0 * D || s1 != s0 * D) atomicAdd(bad, 1u);
    }
}

template <int VAR> double run(float *out, const float *in, const uint32_t *ein, const char *name)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(step_kernel<VAR>, dim3(blocks), dim3(768), 0, 0, out, in, ein, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(step_kernel<VAR>, dim3(blocks), dim3(768), 0, 0, out, in, ein, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / (3.0 * iters);
    printf("%-64s %.3f ms -> %.1f ns per 32-tap wave-step per SIMD\n", name, best, ns);
    return ns;
}

int main()
{
    float *out, *in; uint32_t *ein;
    (void)hipMalloc(&out, 256 * 768 * 4); (void)hipMalloc(&in, 256 * 4); (void)hipMalloc(&ein, 64 * 4);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.5f + i * 0.01f; (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    uint32_t he[64]; for (int i = 0; i < 64; ++i) he[i] = (uint32_t)(i % 41) * 0x01010101u; (void)hipMemcpy(ein, he, sizeof(he), hipMemcpyHostToDevice);
    const double f = run<0>(out, in, ein, "no unpack (floor)");
    const double a = run<1>(out, in, ein, "4 v_cvt_f32_ubyte + 4 v_sub_f32   [kernels until round 3]");
    const double b = run<2>(out, in, ein, "4 byte extractions + 4 v_sub_u32  [integers as fp32 denormals]");
    const double c = run<3>(out, in, ein, "1 packed v_sub_u32 + 8 byte extractions");
    printf("unpack cost per step: cvt form %.1f ns, denormal form %.1f ns, packed-sub form %.1f ns\n", a - f, b - f, c - f);
    // exactness
    const int n = 1 << 20;
    std::vector<float> wl(n), wr(n), acc(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i) {
        wl[i] = (float)std::exp(-rnd() * 30.0); wr[i] = (float)std::exp(-rnd() * 30.0);
        acc[i] = (float)(rnd() * 49000.0 * std::exp(-rnd() * 40.0));
    }
    float *dwl, *dwr, *dacc; uint32_t *dbad;
    (void)hipMalloc(&dwl, n * 4); (void)hipMalloc(&dwr, n * 4); (void)hipMalloc(&dacc, n * 4); (void)hipMalloc(&dbad, 4);
    (void)hipMemcpy(dwl, wl.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dwr, wr.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dacc, acc.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemset(dbad, 0, 4);
    hipLaunchKernelGGL(exact_kernel, dim3(n / 256), dim3(256), 0, 0, dwl, dwr, dacc, dbad, n);
    uint32_t bad = 0; (void)hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    printf("denormal form vs plain form over %d weight pairs x 41 byte values: %u mismatching sums (0 = bit-identical up to the factor 2^-23)\n", n, bad);
    return 0;
}

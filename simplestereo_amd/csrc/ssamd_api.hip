// libssamd.so: host side of the C ABI declared in include/ssamd.h (gfx950 / ROCm).
// Owns device scratch, chooses launch geometry, launches the HIP kernels.
// There is deliberately no CPU code path for the operators in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <list>
#include <map>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ssamd.h"
#include "asw_kernels.hip.h"
#include "asw_pipe_kernel.hip.h"
#include "asw_wave_kernel.hip.h"
#include "asw_wave6_kernel.hip.h"
#include "asw_alt_kernels.hip.h"
#ifndef SSAMD_SINGLE_TU          // (-DSSAMD_SINGLE_TU: everything in this translation unit, as until round 4: tools/build_variants.sh)
namespace ssamd {
#define SSAMD_PIPE_INSTANCE(C, SL, SR, SE) extern template __global__ void asw_aggregate_pipe_kernel<C, SL, SR, SE>(const AswArgs);
#define SSAMD_PIPE_INSTANCE_CG(C, SL, SR, SE) extern template __global__ void asw_aggregate_pipe_kernel<C, SL, SR, SE, true>(const AswArgs);
#define SSAMD_WAVE6_INSTANCE(C, K, CREG) extern template __global__ void asw_aggregate_wave6_kernel<C, K, CREG>(const AswWaveArgs);
#include "asw_instances.inc"
#undef SSAMD_PIPE_INSTANCE
#undef SSAMD_PIPE_INSTANCE_CG
#undef SSAMD_WAVE6_INSTANCE
}  // namespace ssamd
#endif
#include "gsw_kernels.hip.h"
#include "lab_kernels.hip.h"
#include "asw_exact_kernels.hip.h"
#include "rig_kernels.hip.h"

using namespace ssamd;

namespace {

thread_local std::string g_err;
// Locking: every device has its own context and its own mutex (Ctx::mu), so one process can drive several GPUs
// concurrently through the operators (ssamd_*_multi does, one host thread per device).  g_geom_mutex guards the
// small process-wide launch-geometry caches only and is never held across a HIP call.
std::mutex g_geom_mutex;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// Experiment / test hooks.  The SSAMD_* environment variables are read ONCE, when the library is
// loaded; afterwards the table only changes through ssamd_set_option (tests, tools).  The host path of an operator
// call never calls getenv: it reads a thread-local snapshot that is refreshed when the table's version moves.
struct Tuning {
    std::string asw_geom, gsw_geom;   // "XG,DG[,JC[,RX]]" / "XG,DG[,Ty]": forced launch geometry ("" = unset)
    int asw_pipe = -1;                // -1 unset, 0: phase-shifted kernel off, 8 / 16: forced chunk length
    int asw_dephase = -1;             // -1 unset, else the wave order of the phase-shifted kernel
    int asw_evol = 1;                 // 0: in-kernel e tiles instead of the TAD volume
    int asw_wave = -1;                // -1 unset, 0: small-range wave kernel off (any set value bypasses cache and tuner)
    int wave_rx = 0;                  // 0 unset, 8 / 4: forced register tile of the wave kernel
    int wave_wg = 0;                  // 0 unset (one wave per workgroup), 1..4
    int wave_unroll = 1;              // 0: counted build loop
    int wave_merge = 1;               // 0: left and right centres of a strip in separate build rounds (round-2 form)
    int asw_static = 1;               // 0: the phase-shifted kernel always reads its strides from the geometry (round-2 form)
    int evol_max_mb = 0;              // 0 unset; else a cap of the TAD volume in MiB (tests of the paths taken when memory is short)
    int wave_rd = 0;                  // 0: the host decides; 4: never the six-disparities-per-lane form of the wave kernel
    bool no_e2 = false, xor_only = false, multi_allow_repeat = false;
    int alt_queue_cap = 0;            // 0 unset
    int autotune_env = -2;            // -2 unset
    int evol_fail = 0;                // test hook: 1 = the TAD volume's allocation really fails (a hipMalloc no device can serve)
    int lds_relax = 1;                // 0: a phase-shifted tile must fit LDS with its staged colour bytes even when the TAD volume makes them unnecessary
    int wave_creg = 1;                // 0: the wave kernel keeps its window centres in LDS (round-3 form)
    int asw_tail = -1;                // -1: the host decides; 0: never split the last partial round of workgroups into half-width tiles; 1: whenever possible
    int prepass_fuse = 1;             // 0: Lab records and TAD volume as two dependent launches (the form of rounds 2-4)
    int exact_tol = 128;              // fp64 tie-break pass: candidates within this many ulps of the winning cost image are re-evaluated
    int exact_cap = 0;                // 0 unset: queue capacity of the tie-break pass in entries (test hook: a tiny queue overflows)
    int exact_rawcap = 0;             // 0 unset: capacity of the RAW queue of merging calls (test hook)
};
std::mutex g_tune_mutex;
std::atomic<unsigned> g_tune_version{1};

bool tuning_assign(Tuning &t, const std::string &name, const char *v)
{
    auto num = [&](int unset) { return v ? atoi(v) : unset; };
    if (name == "SSAMD_ASW_GEOM") t.asw_geom = v ? v : "";
    else if (name == "SSAMD_GSW_GEOM") t.gsw_geom = v ? v : "";
    else if (name == "SSAMD_ASW_PIPE") t.asw_pipe = num(-1);
    else if (name == "SSAMD_ASW_DEPHASE") t.asw_dephase = num(-1);
    else if (name == "SSAMD_ASW_EVOL") t.asw_evol = num(1);
    else if (name == "SSAMD_ASW_WAVE") t.asw_wave = v ? (atoi(v) != 0 ? 1 : 0) : -1;
    else if (name == "SSAMD_ASW_WAVE_RX") t.wave_rx = num(0);
    else if (name == "SSAMD_ASW_WAVE_WG") t.wave_wg = v ? std::max(1, std::min(4, atoi(v))) : 0;
    else if (name == "SSAMD_ASW_WAVE_UNROLL") t.wave_unroll = num(1);
    else if (name == "SSAMD_ASW_WAVE_MERGE") t.wave_merge = num(1);
    else if (name == "SSAMD_ASW_STATIC") t.asw_static = num(1);
    else if (name == "SSAMD_ASW_EVOL_MAX_MB") t.evol_max_mb = v ? std::max(0, atoi(v)) : 0;
    else if (name == "SSAMD_ASW_WAVE_RD") t.wave_rd = num(0);
    else if (name == "SSAMD_ASW_NO_E2") t.no_e2 = v != nullptr;
    else if (name == "SSAMD_ASW_XOR_ONLY") t.xor_only = v != nullptr;
    else if (name == "SSAMD_MULTI_ALLOW_REPEAT") t.multi_allow_repeat = v != nullptr;
    else if (name == "SSAMD_ALT_QUEUE_CAP") t.alt_queue_cap = v ? std::max(1, atoi(v)) : 0;
    else if (name == "SSAMD_AUTOTUNE") t.autotune_env = v ? (atoi(v) > 0 ? 1 : (atoi(v) < 0 ? -1 : 0)) : -2;
    else if (name == "SSAMD_ASW_EVOL_FAIL") t.evol_fail = num(0);
    else if (name == "SSAMD_ASW_TAIL") t.asw_tail = num(-1);
    else if (name == "SSAMD_ASW_WAVE_CREG") t.wave_creg = num(1);
    else if (name == "SSAMD_ASW_LDS_RELAX") t.lds_relax = num(1);
    else if (name == "SSAMD_ASW_PREPASS_FUSE") t.prepass_fuse = num(1);
    else if (name == "SSAMD_EXACT_TOL") t.exact_tol = v ? std::max(0, atoi(v)) : 128;
    else if (name == "SSAMD_EXACT_CAP") t.exact_cap = v ? std::max(1, atoi(v)) : 0;
    else if (name == "SSAMD_EXACT_RAWCAP") t.exact_rawcap = v ? std::max(1, atoi(v)) : 0;
    else return false;
    return true;
}

const char *const kTuningNames[] = {"SSAMD_ASW_GEOM", "SSAMD_GSW_GEOM", "SSAMD_ASW_PIPE", "SSAMD_ASW_DEPHASE", "SSAMD_ASW_EVOL",
                                    "SSAMD_ASW_WAVE", "SSAMD_ASW_WAVE_RX", "SSAMD_ASW_WAVE_WG", "SSAMD_ASW_WAVE_UNROLL",
                                    "SSAMD_ASW_WAVE_MERGE", "SSAMD_ASW_STATIC", "SSAMD_ASW_EVOL_MAX_MB", "SSAMD_ASW_WAVE_RD", "SSAMD_ASW_NO_E2", "SSAMD_ASW_XOR_ONLY", "SSAMD_MULTI_ALLOW_REPEAT",
                                    "SSAMD_ALT_QUEUE_CAP", "SSAMD_AUTOTUNE", "SSAMD_ASW_EVOL_FAIL", "SSAMD_ASW_TAIL", "SSAMD_ASW_WAVE_CREG", "SSAMD_ASW_LDS_RELAX",
                                    "SSAMD_EXACT_TOL", "SSAMD_EXACT_CAP", "SSAMD_EXACT_RAWCAP", "SSAMD_ASW_PREPASS_FUSE"};

std::map<std::string, std::string> g_tuning_env;      // what the process was started with: ssamd_set_option(name, NULL) goes back to THIS
Tuning tuning_from_env()
{
    Tuning t;
    for (const char *n : kTuningNames)
        if (const char *v = getenv(n)) {      // the only getenv calls of the library
            g_tuning_env[n] = v;
            (void)tuning_assign(t, n, v);
        }
    return t;
}
Tuning g_tuning = tuning_from_env();

const Tuning &tune()
{
    thread_local Tuning snap;
    thread_local unsigned seen = 0;
    const unsigned v = g_tune_version.load(std::memory_order_acquire);
    if (v != seen) {
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        snap = g_tuning;
        seen = v;
    }
    return snap;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? SSAMD_ENOMEM : SSAMD_EHIP, "%s failed: %s",    \
                        #expr, hipGetErrorString(e_));                                             \
    } while (0)

// ------------------------------------------------------------------ scratch
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SSAMD_OK;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&ptr, want);
        if (e != hipSuccess) {
            ptr = nullptr;
            (void)hipGetLastError();      // ROCm 7 keeps the failure as the thread's last error: a caller that falls back must not see it again
            return fail(SSAMD_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return SSAMD_OK;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
    }
};

struct Profile {
    bool on = false;
    struct Pair { hipEvent_t a, b; int slot; };
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    double ms[SSAMD_K_COUNT] = {0};
    long long n[SSAMD_K_COUNT] = {0};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void drain()
    {
        for (auto &p : pending) {
            float t = 0.f;
            if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
                ms[p.slot] += t;
                n[p.slot] += 1;
            }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
};

// Small parameter-keyed device tables (ASW proximity weights per (winSize, gammaP), GSW weight table per gamma):
// a matcher that alternates between parameter sets finds its table again instead of re-uploading it behind a
// stream synchronisation.  An entry owns its host copy, so the upload is an ordinary asynchronous copy on the
// calling stream; later calls on other streams are ordered behind it by ScratchOrder.
struct TableEntry {
    int k0 = 0; double k1 = 0;
    DevBuf dev;
    std::vector<float> host;
    std::vector<double> host64;          // (the fp64 proximity table of the tie-break pass)
};
struct TableCache {
    std::list<TableEntry> entries;       // most recently used first
    size_t max_entries;
    explicit TableCache(size_t n) : max_entries(n) {}
    TableEntry *find(int k0, double k1)
    {
        for (auto it = entries.begin(); it != entries.end(); ++it)
            if (it->k0 == k0 && it->k1 == k1) {
                entries.splice(entries.begin(), entries, it);
                return &entries.front();
            }
        return nullptr;
    }
};

struct Ctx {
    std::mutex mu;                      // serialises the calls on this device
    int dev = -1;
    int cus = 256;                      // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    bool lut_ready = false;
    hipStream_t stream = nullptr;       // used by the host-buffer entry points
    DevBuf imgL, imgR, recL, recR, keyL, keyR, disp, costs, lab, altq, evol, altdisp;
    DevBuf xlabL, xlabR, xflags, xqueue, xcost, xslots, xctr, xraw, xwtab;      // fp64 tie-break pass (asw_exact_kernels.hip.h)
    unsigned int xcap = 0, xrawcap = 0; // queue capacities of the last exact call
    TableCache proxTabs{8}, gswTabs{4}, proxTabs64{8};
    std::map<const void *, int> max_dyn_lds;   // hipFuncAttributeMaxDynamicSharedMemorySize already granted per kernel
    hipEvent_t scratch_free = nullptr;  // recorded after the last kernel that uses the scratch buffers
    int evol_small_calls = 0;           // consecutive calls that needed less than a quarter of the TAD volume's capacity
    long long evol_fallbacks = 0;       // calls that ran without the volume (in-kernel e tiles) or off the wave kernel for lack of memory
    long long tail_splits = 0;          // phase-shifted launches whose last partial round of workgroups ran as half-width tiles
    long long exact_calls = 0;          // ASW calls that ran the fp64 tie-break pass
    long long static_tile_mismatch = 0; // pipe launches whose strides named a static tile that the full geometry did not match
    Profile prof;
};

Ctx g_ctx[16];

// The calling thread's current HIP device is restored when an entry point returns: an operator asked to run on
// device k must not leave the caller's later allocations or launches on device k.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// hipFuncSetAttribute costs a runtime round trip: ask only when a launch needs more dynamic LDS than the kernel
// has been granted on this device so far
int grant_dyn_lds(Ctx &c, const void *kernel, int bytes);

struct Timed {   // brackets one kernel launch with events when profiling is on
    Ctx &c; hipStream_t s; int slot; hipEvent_t a = nullptr, b = nullptr;
    Timed(Ctx &c_, hipStream_t s_, int slot_) : c(c_), s(s_), slot(slot_)
    {
        if (c.prof.on) { a = c.prof.get(); b = c.prof.get(); (void)hipEventRecord(a, s); }
    }
    ~Timed()
    {
        if (a) { (void)hipEventRecord(b, s); c.prof.pending.push_back({a, b, slot}); }
    }
};

// A locked device context: makes `device` (-1: the calling thread's current one) current for the duration of the
// entry point, takes that device's mutex and restores the caller's device afterwards.
struct CtxLock {
    DeviceGuard guard;                   // destroyed last: restores the caller's device after the unlock
    std::unique_lock<std::mutex> lk;
    Ctx *c = nullptr;
    Ctx *operator->() { return c; }
    Ctx &operator*() { return *c; }
};

int get_ctx(int device, CtxLock &out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SSAMD_ENODEVICE, "no HIP device visible: libssamd has no CPU fallback");
    if (device < 0) device = out.guard.saved >= 0 ? out.guard.saved : 0;
    if (device >= n || device >= 16) return fail(SSAMD_EINVAL, "device ordinal %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    Ctx &c = g_ctx[device];
    out.lk = std::unique_lock<std::mutex>(c.mu);
    out.c = &c;
    if (c.dev < 0) {
        c.dev = device;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c.cus = cus;
        else (void)hipGetLastError();
        HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c.scratch_free, hipEventDisableTiming));
    }
    if (!c.lut_ready) {
        // sRGB byte -> linear*100 in the reference's float arithmetic (colorconversion.hpp:19-37); powf = glibc's algorithm restated
        // (glibc_math.hip.h, host side), NOT the host's libm: the same 256 floats on any host (oracle/libm_check.c proves them equal
        // to glibc's, tests/test_gpu_libm_independence.py runs the path with the process's exp / powf replaced by garbage)
        float lut[256];
        for (int v = 0; v < 256; ++v) {
            float x = v / 255.0;
            if (x > 0.04045) x = glibc_powf_pos((float)((x + 0.055) / 1.055), (float)2.4);
            else x /= 12.92;
            x *= 100;
            lut[v] = x;
        }
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_lin100), lut, sizeof(lut)));
        c.lut_ready = true;
    }
    return SSAMD_OK;
}

int grant_dyn_lds(Ctx &c, const void *kernel, int bytes)
{
    int &granted = c.max_dyn_lds[kernel];
    if (bytes <= granted || bytes <= 48 * 1024) return SSAMD_OK;     // 48 KiB need no opt-in
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    granted = bytes;
    return SSAMD_OK;
}

// The scratch buffers (pixel records, WTA keys, tables) are shared by every call on a device.  Calls are
// serialised on the host by the device's mutex; across streams the next call's stream waits for the previous call's
// last kernel, so callers may use any stream without synchronising between operators.
struct ScratchOrder {
    Ctx &c; hipStream_t s;
    ScratchOrder(Ctx &c_, hipStream_t s_) : c(c_), s(s_) { (void)hipStreamWaitEvent(s, c.scratch_free, 0); }
    ~ScratchOrder() { (void)hipEventRecord(c.scratch_free, s); }
};

int check_common(int H, int W, int win, int minD, int maxD, int row0, int rows)
{
    if (H <= 0 || W <= 0) return fail(SSAMD_EINVAL, "Wrong image dimensions!");
    if (!(win > 0 && win % 2 == 1)) return fail(SSAMD_EINVAL, "winSize must be a positive odd number!");
    if (minD < 0) return fail(SSAMD_EINVAL, "minDisparity must be >= 0 (negative values are undefined behaviour in the reference)");
    if (W > 32767 || maxD > 32767) return fail(SSAMD_ELIMIT, "width / maxDisparity exceed the int16 disparity range");
    if (row0 < 0 || rows < 0 || row0 + rows > H) return fail(SSAMD_EINVAL, "output row range [%d,%d) outside the image (height %d)", row0, row0 + rows, H);
    if (win > 255) return fail(SSAMD_ELIMIT, "winSize %d > 255 not supported", win);
    if (rows > 65535) return fail(SSAMD_ELIMIT, "more than 65535 output rows per call: split the image into row strips");
    return SSAMD_OK;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------ ASW geometry
thread_local bool t_pipe_full_lds = false;      // true while a geometry is planned for a call that has no TAD volume

bool asw_layout_e(AswGeom &g, int win, int XG, int DG, size_t limit, int JC, int Rx, bool e2, bool odd_pitch = false,
                  bool pipe = false)
{
    g.Rx = Rx;
    g.JC = JC >= win ? win : JC;                 // tap columns staged per chunk; win = the whole row at once
    g.pipe = 0; g.NC = 1; g.JCmax = g.JC; g.dephase = 0; g.wave_rx = 0;
    if (pipe) {
        // phase-shifted kernel (asw_pipe_kernel.hip.h): chunks start at multiples of JC (8 or 16), a tail shorter than
        // the 8-column register tile is merged into the last chunk; needs >= 2 chunks, two e tiles, the 8-column tile
        // chunk starts are multiples of JC (itself a multiple of the register tile's columns); the chunk count is
        // win / JC rounded, the last chunk takes what is left (win 35: JC 16 -> 16, 19; JC 8 -> 8, 8, 8, 11; JC 12 -> 12, 12, 11)
        if (Rx != 8 || JC % Rx || !e2) return false;
        g.NC = (win + JC / 2) / JC;
        if (g.NC < 2 || (g.NC - 1) * JC >= win) return false;
        g.pipe = 1;
        // waves 0-3 build before they aggregate (see the kernel): pays with three or four waves per SIMD (12-wave
        // groups: 1080p/193 41.8 -> 41.0 ms), costs with two (640x480/65, 8 waves: 3.11 -> 3.26 ms)
        g.dephase = tune().asw_dephase >= 0 ? tune().asw_dephase : (round_up(XG * DG, 64) / 64 >= 12 ? 1 : 0);
        g.JCmax = std::max(JC, win - (g.NC - 1) * JC);
    }
    const int wrows = g.pipe ? 2 * g.JCmax : (g.JC < win ? 2 * g.JC : win);   // chunk buffers alternate
    const int wcols = g.JC;                      // tap columns a weight-build pass covers

    const int p = win / 2;
    g.XG = XG; g.DG = DG;
    g.Tx = Rx * XG; g.Dc = ASW_RD * DG;
    g.threads = round_up(XG * DG, 64);
    g.nL = g.Tx + 2 * p;
    g.nRc = g.Tx + g.Dc - 1;
    g.nR = g.nRc + 2 * p;
    // parity-split rows (asw_split_pos): two halves of ceil(n/8)*4 floats; +1 block so that the halves
    // start on different banks phases and reads one block past the end stay inside the row
    g.hL = ((g.Tx + 7) / 8) * 4 + 4;
    g.SL = 2 * g.hL;
    g.hR = ((g.nRc + 4 + 7) / 8) * 4 + 4;
    g.SR = 2 * g.hR;
    int P = 8;                                  // dword slots per e row: closed under XOR with emask
    while (P < DG && P < 32) P <<= 1;           //   power of two up to 32, then multiples of 32
    if (P < DG) P = round_up(DG, 32);
    g.Se = 4 * P;
    g.emask = std::min(P, 32) - 1;
    if (odd_pitch) {                            // plain rows with an odd dword pitch instead of the XOR swizzle
        g.Se = 4 * (DG | 1);
        g.emask = 0;
    }
    if (g.pipe) {
        // plain rows, lanes along the disparity groups (asw_pipe_kernel.hip.h): a thread reads floats
        // [8 xg, 8 xg + 8) of a wL row and [8 xg - 4 dg + Dc - 4, + 12) of a wR row (the last one is index nRc, unused)
        g.hL = g.hR = 0;
        g.SL = round_up(g.Tx, 4);
        g.SR = round_up(g.nRc + 1, 4);
        // e rows: one dword per disparity group, pitch a multiple of 16 bytes so that a tile is an aligned contiguous
        // block of the pre-computed volume (LDS-DMA moves 16 bytes per lane)
        g.Se = 16 * ((DG + 3) / 4);
        g.emask = 0;
    }
    // weight build balance: (centres x segments) tasks over the workgroup's threads
    {
        const int ncen = g.Tx + g.nRc;
        int best_cost = 1 << 30;
        for (int ns = 1; ns <= wcols && ns <= 8; ++ns) {
            const int len = (wcols + ns - 1) / ns, rounds = (ncen * ns + g.threads - 1) / g.threads;
            const int cost = rounds * (round_up(len, ASW_WB) + 2);       // evaluated in batches of ASW_WB
            if (cost < best_cost) { best_cost = cost; g.wseg = ns; g.wlen = len; }
        }
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return (int)o; };
    g.off_wL = take((size_t)wrows * g.SL * 4);
    g.off_wR = take((size_t)wrows * g.SR * 4);
    g.e_bytes = (int)(((size_t)g.nL * g.Se + 15) & ~(size_t)15);
    g.e2 = (e2 && g.JC < win && (win + g.JC - 1) / g.JC >= 2) ? 1 : 0;
    if (g.pipe && !g.e2) return false;
    g.off_e = take((size_t)g.e_bytes * (g.e2 ? 2 : 1));
    g.off_labL = take((size_t)g.nL * 16 * 2);    // staging is double-buffered (prefetch of the next row)
    g.off_labR = take((size_t)g.nR * 16 * 2);
    if (!g.pipe) {
        g.off_bgrL = take((size_t)g.nL * 4 * 2);
        g.off_bgrR = take((size_t)g.nR * 4 * 2);
    }
    g.off_bestL = take((size_t)g.Tx * 8);
    g.off_bestR = take((size_t)(g.nRc + 1) * 8);
    g.off_cen = take((size_t)(g.Tx + g.nRc) * 16);
    g.off_prox = take((size_t)win * 4 * 2);      // one window row of proximity weights, double-buffered
    g.lds_bytes_evol = (int)off;
    if (g.pipe) {
        // the staged colour bytes only feed the in-kernel e tiles: LAST in the layout, so that a launch that has the pre-computed
        // TAD volume asks for lds_bytes_evol and leaves them out (round 4: LDS is what bounds the resident workgroups of mid-size tiles)
        g.off_bgrL = take((size_t)g.nL * 4 * 2);
        g.off_bgrR = take((size_t)g.nR * 4 * 2);
    }
    g.lds_bytes = (int)off;
    // the phase-shifted kernel normally runs with the TAD volume and then does not allocate the staged colour bytes: a tile may
    // count on that (round 4); a call that cannot have the volume re-plans with t_pipe_full_lds set (asw_device_impl)
    if (g.pipe && tune().asw_evol != 0 && tune().lds_relax != 0 && !t_pipe_full_lds) return (size_t)g.lds_bytes_evol <= limit;
    return off <= limit;
}

// Chunked geometries first try two e tiles (no row-start barrier, asw_kernels.hip.h); when that does not fit the
// LDS budget they fall back to one.
bool asw_layout(AswGeom &g, int win, int XG, int DG, size_t limit, int JC = 1 << 20, int Rx = ASW_RX, bool odd_pitch = false)
{
    if (!tune().no_e2 && asw_layout_e(g, win, XG, DG, limit, JC, Rx, true, odd_pitch) && g.e2) return true;
    return asw_layout_e(g, win, XG, DG, limit, JC, Rx, false, odd_pitch);
}

// Average number of LDS passes of the aggregation loop's e-row read (one dword per lane; a wave is served in two
// halves of 32 lanes, a pass per distinct address that shares a bank) for an e layout: lanes = consecutive thread
// ids, thread (xg, dg) reads dword dg (XOR-swizzled with row / Rx & emask) of row Rx*xg + n.
double asw_e_read_passes(const AswGeom &g)
{
    const int P = g.Se / 4, T = g.XG * g.DG;
    long long tot = 0, cnt = 0;
    for (int n = 0; n < g.Rx; ++n)
        for (int base = 0; base < T; base += 32) {
            int hits[64] = {0}, worst = 0;
            for (int l = 0; l < 32 && base + l < T; ++l) {
                const int tid = base + l, xg = tid % g.XG, dg = tid / g.XG, ul = g.Rx * xg + n;
                worst = std::max(worst, ++hits[(ul * P + (dg ^ ((ul / g.Rx) & g.emask))) & 63]);
            }
            tot += worst; ++cnt;
        }
    return cnt ? (double)tot / cnt : 1.0;
}

// The e-tile scheme is decided for the chosen tile only (the search prices LDS with the swizzled form): rows with an
// odd dword pitch are smaller (DG|1 instead of a power of two / multiple of 32 dwords) and often conflict less for
// narrow thread grids; the XOR swizzle wins for wide ones.  Take the odd pitch when it makes room for a second e
// tile, or when it does not read slower.
void asw_pick_e_scheme(AswGeom &g, int win)
{
    if (tune().xor_only) return;
    AswGeom alt;
    if (!asw_layout(alt, win, g.XG, g.DG, 160 * 1024, g.JC >= win ? (1 << 20) : g.JC, g.Rx, true)) return;
    // two e tiles (one barrier less per window row: 1080p/193 45.96 -> 44.7 ms) outweigh a few bank conflicts of a
    // one-dword read; among equals the layout with fewer passes wins
    const bool take = alt.e2 != g.e2 ? alt.e2 > g.e2 : asw_e_read_passes(alt) <= asw_e_read_passes(g) + 1e-9;
    if (take) {
        alt.nchunks = g.nchunks;
        g = alt;
    }
}

// Phase-shifted kernel for a chosen tile (asw_pipe_kernel.hip.h): same XG x DG thread grid and register tile, tap
// columns in chunks of 8 (or 16) with the tail merged, two e tiles.  Taken whenever it fits (8-column tile, window of
// at least two chunks, LDS); the sums and their order are those of asw_aggregate_kernel, so maps do not change.
// SSAMD_ASW_PIPE=0 disables it, =8 / =16 force the chunk length (experiments and tests).
void asw_try_pipe(AswGeom &g, int win)
{
    const int want = tune().asw_pipe;
    if (want == 0 || g.Rx != 8) return;
    for (int JC : {16, 8}) {
        if (want > 0 && JC != want) continue;
        // chunks of 8 double the barriers per window row: measured to pay only with three waves per SIMD
        // (4096x2160/257: 265 -> 245 ms, 1080p/129/win 21: 12.7 -> 11.3 ms; 8-wave tiles lose 5-15 %)
        if (want < 0 && JC == 8 && round_up(g.XG * g.DG, 64) / 64 < 12) continue;
        AswGeom alt;
        if (!asw_layout_e(alt, win, g.XG, g.DG, 160 * 1024, JC, 8, true, false, true)) continue;
        alt.nchunks = g.nchunks;
        g = alt;
        return;
    }
}

// Wave-autonomous kernel for small disparity ranges (asw_wave_kernel.hip.h): geometry of one wave's strip and its
// slice of LDS.  false: the range does not fit one chunk of at most ASW_WAVE_MAX_DG disparity groups.
static constexpr int ASW_WAVE_MAX_DG = 16;
// One candidate strip: nxg column groups, left / right centres in separate build rounds or merged into one list.
bool asw_wave_layout_one(AswWaveGeom &g, int win, int DG, int rx, int nxg, bool merged, int rd = ASW_RD, bool creg = false)
{
    g.RX = rx;
    g.RD = rd;
    g.creg = 0;
    const int p = win / 2;
    g.DG = DG;
    g.NXG = nxg;
    g.Txw = rx * g.NXG;
    g.Dc = rd * g.DG;
    g.lanes = g.NXG * g.DG;
    g.nLw = g.Txw + 2 * p;
    g.nRcw = g.Txw + g.Dc - 1;
    g.nRw = g.nRcw + 2 * p;
    g.merged = merged ? 1 : 0;
    if (merged) {
        // one list of Txw + nRcw centres: the right weights follow the left ones directly, the row is padded to whole rounds
        g.K = (g.Txw + g.nRcw + 63) / 64;
        g.SLw = g.Txw;
        g.SRw = round_up(g.Txw + g.nRcw + 1, 64) - g.Txw;
    } else {
        g.K = (g.Txw + 63) / 64 + (g.nRcw + 63) / 64;
        g.SLw = round_up(g.Txw, 64);                   // weight rows padded to whole 64-lane build rounds
        g.SRw = round_up(g.nRcw + 1, 64);
    }
    // bytes per e column: an odd number of dwords, so that the e dwords the lanes of a wave read in one step (column
    // group stride rx * Se) spread over the LDS banks -- with Se = 32 the 12 column groups of D 0..16 all hit the same
    // five banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.21)
    g.Se = rd == 6 ? 8 * ((g.DG + 1) | 1) : 4 * (g.DG | 1);        // (six per lane: 8-byte slots, an odd number of them and one to spare)
    g.waves = tune().wave_wg ? tune().wave_wg : 1;
    // order matters: the build's last round reads up to 127 entries past the end of the centres and of each pixel
    // row (asw_wave_kernel.hip.h) -- into the array that follows, never past the e tile -- and the merged build
    // relies on pixR starting right behind pixL
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return (int)o; };
    g.off_w = take((size_t)(g.SLw + g.SRw) * 4 * 2);        // two rows: tap columns j and j + 1
    // centres in registers (asw_aggregate_wave_kernel<..., CREG>): four-per-lane, 4-column tile, merged build of at most four rounds
    g.creg = creg && merged && rx == 4 && g.K >= 2 && (rd == 6 ? g.K <= 3 : g.K <= 4) ? 1 : 0;      // (the instantiations that exist)
    g.off_cen = take(g.creg ? 0 : (size_t)(g.Txw + g.nRcw) * 16);
    g.off_pixL = take((size_t)g.nLw * 16);
    g.off_pixR = take((size_t)g.nRw * 16);
    // (+ rx e columns of slack: the lanes past the last column group read inside the slice.  The six-per-lane kernel clamps
    //  their column group instead, round 4: LDS is granted in 512-byte granules, and at win 35 those 160 bytes decide whether
    //  11 or 12 of its waves are resident per CU -- 13 424 -> 13 264 bytes, 6.06 -> 5.8 ms at 1080p / D 0..16)
    g.off_e = take(std::max((size_t)g.nLw * g.Se, (size_t)(129 + 2 * p) * 16) + (rd == 6 ? 0 : (size_t)rx * g.Se));
    // the winner arrays are only used after the last window row: they share the pixel rows' space
    g.off_bestL = g.off_pixL;
    g.off_bestR = g.off_pixL + (int)(((size_t)g.Txw * 8 + 15) & ~(size_t)15);
    if ((size_t)g.off_bestR + (size_t)(g.nRcw + 1) * 8 > off) off = (size_t)g.off_bestR + (size_t)(g.nRcw + 1) * 8;
    g.wave_lds = (int)((off + 15) & ~(size_t)15);
    return g.off_pixR == g.off_pixL + g.nLw * 16 && (size_t)g.wave_lds * g.waves <= 160 * 1024;
}

// The strip of a wave: DG disparity groups x NXG <= 64 / DG column groups.  Round 3: the number of column groups and
// whether the left and right centres are built as one list are chosen by the cost per column of a tap column's work,
// K build rounds (~17 issue slots each: one weight per lane) + the taps (~59 slots with the 4-column tile, ~110 with
// the 8-column one).  SSAMD_ASW_WAVE_MERGE=0 restores the round-2 form (all column groups, separate rounds).
// rx: 8 or 4 columns per lane; 4 | 16 (= 20, an autotuning candidate, AswGeom::wave_rx): 4 columns and never six disparities per lane
bool asw_wave_layout(AswWaveGeom &g, int win, int nD, int rx, bool creg = false)
{
    const bool never6 = (rx & 16) != 0;
    rx &= 15;
    creg = creg && tune().wave_creg != 0;
    const int DG = (nD + ASW_RD - 1) / ASW_RD;
    if (DG < 1 || DG > ASW_WAVE_MAX_DG) return false;
    const int nxg_max = 64 / DG;
    if (tune().wave_merge == 0) return asw_wave_layout_one(g, win, DG, rx, nxg_max, false);
    const double c_round = 17.0, c_taps = rx == 8 ? 110.0 : 59.0;
    double best = 1e30;
    bool found = false;
    // (separate rounds first: on a tie they win -- measured 1.5 % faster at D 0..32, where both forms take three rounds;
    //  the merged form only where its straight-line instantiations exist, K <= 4: the counted loop with its per-round
    //  select lost 7 % at D 0..3, eight rounds instead of nine)
    for (int nxg = nxg_max; nxg >= std::max(1, nxg_max - 4); --nxg)
        for (int merged = 0; merged <= 1; ++merged) {
            AswWaveGeom c;
            if (merged && nxg != nxg_max && tune().wave_merge == 2) continue;
            if (!asw_wave_layout_one(c, win, DG, rx, nxg, merged != 0, ASW_RD, creg)) continue;
            if (merged && c.K > 4) continue;
            if (!merged && nxg != nxg_max) continue;     // fewer column groups only pay through a saved merged round
            const double cost = (c.K * c_round + c_taps) / (double)c.Txw;
            if (cost < best - 1e-9) { best = cost; g = c; found = true; }
        }
    // Six disparities per lane (asw_wave6_kernel.hip.h, 4-column tile, merged rounds only): where the range pads badly to groups of
    // four -- 17 and 18 disparities, the class default among them: three groups of six, 21 column groups, three build rounds
    // for 84 columns -- it must beat the four-per-lane strip by 5 % of the modelled cost to be taken
    if (found && rx == 4 && !never6 && tune().wave_rd != 4 && tune().wave_merge != 0) {
        const int DG6 = (nD + 5) / 6;
        if (DG6 >= 1 && DG6 <= 10) {
            const int nxg6 = 64 / DG6;
            for (int nxg = nxg6; nxg >= std::max(1, nxg6 - 4); --nxg) {
                AswWaveGeom c;
                if (!asw_wave_layout_one(c, win, DG6, 4, nxg, true, 6, creg) || c.K > 4) continue;
                const double cost = (c.K * c_round + 87.0) / (double)c.Txw;
                if (cost < 0.95 * best) { best = cost / 0.95; g = c; }
            }
        }
    }
    return found;
}

// Which wave kernel (0: none) serves a window / disparity range.  Measured on 1080p and VGA frames, windows 11..35
// (profiles/r02_wave_sweep.txt): the wave kernel beats the workgroup kernels up to 48 disparities; the 4-column tile
// (more waves per SIMD, half the LDS per wave) wins up to 16 disparities, the 8-column tile above.
// Round 3: with the merged build rounds (two rounds for the 32 + 83..95 centres of a four-column-group strip) the wave
// kernel also wins for 49..64 disparities -- 14.0-14.1 ms against 15.9-16.6 ms at 1080p / win 35, 2.64 vs 3.09 ms at VGA
// (profiles/r03_wave_range_49_64_ab.txt); from 65 disparities (three column groups per wave) the phase-shifted kernel is ahead
// again (17.2 vs 18.7 ms at D 0..64), so the limit is 16 disparity groups.
// SSAMD_ASW_WAVE=0 disables it, SSAMD_ASW_WAVE_RX=8|4 forces a tile (experiments / tests); SSAMD_ASW_EVOL=0 (in-kernel e
// tiles) also disables it, the wave kernel needs the TAD volume.
static constexpr int ASW_WAVE_MAX_ND = 64;
int asw_wave_pick(int win, int nD)
{
    if (tune().asw_wave == 0 || tune().asw_evol == 0) return 0;
    if (nD < 1 || nD > ASW_WAVE_MAX_ND || win > 63) return 0;
    AswWaveGeom wg;
    if (const int rx = tune().wave_rx) {
        return (rx == 8 || rx == 4) && asw_wave_layout(wg, win, nD, rx) ? rx : 0;
    }
    // (round 3, merged build rounds: five disparity groups -- 17..20 disparities, the class default among them -- build
    //  48 + 67 centres in two rounds with the 4-column tile: 6.10 vs 6.27 ms at 1080p / D 0..16; from six groups on the
    //  8-column tile wins, 6.69 vs 7.63 ms at D 0..20)
    const int first = nD <= 20 ? 4 : 8, second = 12 - first;
    if (asw_wave_layout(wg, win, nD, first)) return first;
    return asw_wave_layout(wg, win, nD, second) ? second : 0;
}

// Pick the workgroup tile (XG column groups x DG disparity groups, nchunks disparity chunks)
// with an occupancy-aware cost model calibrated on MI355X (profiles/r01_*):
//   - the kernel needs 168 VGPRs -> 3 waves per SIMD; a workgroup of w waves puts ceil(w/4)
//     on each SIMD, so k = min(floor(3 / ceil(w/4)), floor(160 KiB / LDS)) workgroups are
//     resident per CU.  Measured: 2 x 6-wave groups do NOT co-reside (87 ms), one 12-wave
//     group does (55 ms) on the 1080p/193/35 workload.
//   - per window row a thread spends M cycles aggregating and B cycles building weights / e
//     tiles; B shrinks with the tile (fewer window centres per (x,d) pair).
//   - padding of the disparity range, idle lanes, partial x tiles and the last partial wave of
//     workgroups over the 256 CUs are charged as lost throughput.
int asw_search_geometry(AswGeom &best, int W, int rows, int win, int nD, std::vector<AswGeom> *shortlist = nullptr);

// The search walks a few thousand candidate tiles (0.1-0.3 ms on the host): remember the answer per problem shape,
// a video stream asks the same question every frame.  (Both maps are guarded by g_geom_mutex.)
std::map<std::array<int, 4>, AswGeom> g_asw_geom_cache;
std::map<std::array<int, 4>, bool> g_asw_geom_tuned;      // shapes whose cached geometry was picked by measurement
// autotuning mode: 1 always, 0 never, -1 (default) only for small problems, where the ~50 trial launches cost
// at most about 0.4 s once and where the cost model is least reliable
std::atomic<int> g_autotune{g_tuning.autotune_env != -2 ? g_tuning.autotune_env : -1};
constexpr double ASW_AUTOTUNE_SMALL_TAPS = 6.0e10;      // window taps per call (about 6-8 ms of kernel time; 3e10 until round 4)

// experiment / test hooks that force a kernel form: such calls neither read nor write the geometry cache and are not autotuned
bool asw_geometry_forced()
{
    const Tuning &t = tune();
    return !t.asw_geom.empty() || t.asw_wave >= 0 || t.wave_rx != 0 || t.wave_merge != 1 || t.asw_pipe >= 0 || t.asw_dephase >= 0 ||
           t.asw_evol != 1 || t.wave_wg != 0 || t.no_e2 || t.xor_only || t.asw_static != 1 || t.evol_max_mb != 0 || t.wave_rd != 0 ||
           t.wave_creg != 1 || t.lds_relax != 1;
}

int asw_choose_geometry(AswGeom &best, int W, int rows, int win, int nD)
{
    if (asw_geometry_forced()) return asw_search_geometry(best, W, rows, win, nD);          // tuning hooks: never cached
    std::lock_guard<std::mutex> glk(g_geom_mutex);
    const std::array<int, 4> key{W, rows, win, nD};
    auto it = g_asw_geom_cache.find(key);
    if (it != g_asw_geom_cache.end()) { best = it->second; return SSAMD_OK; }
    const int rc = asw_search_geometry(best, W, rows, win, nD);
    if (rc == SSAMD_OK) {
        if (g_asw_geom_cache.size() > 256) { g_asw_geom_cache.clear(); g_asw_geom_tuned.clear(); }
        g_asw_geom_cache[key] = best;
    }
    return rc;
}

// shortlist (autotuning): the best-scoring geometry of every structurally different class of candidates
// (register tile, tap-column chunking, disparity chunks, waves per group), best classes first
int asw_search_geometry(AswGeom &best, int W, int rows, int win, int nD, std::vector<AswGeom> *shortlist)
{
    std::map<std::array<int, 4>, std::pair<double, AswGeom>> classes;
    // tuning hook: SSAMD_ASW_GEOM="XG,DG[,JC[,RX]]" forces the tile shape (experiments and tests only)
    if (!tune().asw_geom.empty()) {
        int XG = 0, DG = 0, JCe = 1 << 20, Rx = ASW_RX;
        if (sscanf(tune().asw_geom.c_str(), "%d,%d,%d,%d", &XG, &DG, &JCe, &Rx) >= 2 && XG > 0 && DG > 0 && XG * DG <= ASW_MAX_THREADS &&
            (Rx == 8 || Rx == 4)) {
            if (JCe <= 0 || JCe % Rx) JCe = 1 << 20;
            if (!asw_layout(best, win, XG, DG, 160 * 1024, JCe, Rx)) return fail(SSAMD_ELIMIT, "SSAMD_ASW_GEOM does not fit LDS");
            best.nchunks = (nD + best.Dc - 1) / best.Dc;
            asw_pick_e_scheme(best, win);
            asw_try_pipe(best, win);
            return SSAMD_OK;
        }
    }
    const double c_tap = 10.9, c_w = 70.0, c_e = 60.0, c_stage = 40.0;   // cycles (one SIMD lane-slot)
    double best_score = -1.0;
    bool found = false;
    for (int nch = 1; nch <= nD; ++nch) {
        const int per = (nD + nch - 1) / nch;
        const int DG = round_up(per, ASW_RD) / ASW_RD;
        if (DG > 128) continue;
        if ((nD + DG * ASW_RD - 1) / (DG * ASW_RD) != nch) continue;
        // register tile 8x4 (168 VGPRs: 3 waves per SIMD), or 4x4 (<= 128 VGPRs: 4 waves per SIMD, twice the
        // threads per tile column) for small disparity ranges, where LDS capacity bounds the resident waves
        // (measured, 1080p / win 35: D 0..16 16.7 -> 10.5 ms, D 0..32 14.3 -> 13.2 ms, D 0..47 17.2 -> 14.7 ms,
        //  D 0..64 no gain)
        for (int Rx : {8, 4}) {
        if (Rx == 4 && nD > 56) continue;
        const int max_wps = Rx == 8 ? 3 : 4;
        const int xg_cap = std::min(ASW_MAX_THREADS / DG, (W + Rx - 1) / Rx);
        const int pipe_env = tune().asw_pipe;
        for (int XG = xg_cap; XG >= 1; --XG)
        for (int cand = 0; cand < 6; ++cand) {
            // candidates 0-3: asw_aggregate_kernel with whole window rows or tap-column chunks of 16 / 8 / 4;
            // candidates 4-5: the phase-shifted kernel (8-column tile) with chunks of 16 / 8
            static const int jcs[6] = {1 << 20, 16, 8, 4, 16, 8};
            const int JC = jcs[cand];
            const bool piped = cand >= 4;
            AswGeom g;
            if (piped) {
                if (Rx != 8 || pipe_env == 0 || (pipe_env > 0 && pipe_env != JC)) continue;
                if (pipe_env < 0 && JC == 8 && round_up(XG * DG, 64) / 64 < 12) continue;      // see asw_try_pipe
                if (!asw_layout_e(g, win, XG, DG, 160 * 1024, JC, 8, true, false, true)) continue;
            } else {
                if (JC < (1 << 20) && (JC >= win || JC % Rx)) continue;
                if (!asw_layout(g, win, XG, DG, 160 * 1024, JC, Rx)) continue;
            }
            g.nchunks = nch;
            const int waves = g.threads / 64, per_simd = (waves + 3) / 4;
            // (the phase-shifted kernel normally runs with the TAD volume and then leaves the staged colour bytes out of its LDS)
            const int k = std::min(max_wps / per_simd, (160 * 1024) / (piped && tune().asw_evol != 0 ? g.lds_bytes_evol : g.lds_bytes));
            if (k < 1) continue;
            // per-thread aggregation cycles of one window row; the 4-column tile spends the same address and
            // e-row work on half the taps; the phase-shifted kernel's step is 107 instead of 111 instructions
            // with a third of the bank conflicts
            const double M = (double)win * Rx * ASW_RD * (Rx == 8 ? (piped ? 0.93 * c_tap : c_tap) : c_tap * 1.15);
            const int ncen = g.Tx + g.nRc;
            const int njc = piped ? g.NC : (win + g.JC - 1) / g.JC;     // weight-build passes (= barriers) per window row
            double B;
            if (piped)      // no e tiles (TAD volume), one centre per thread, the build partly under other waves' taps
                B = (double)ncen * win / g.threads * 28.0 + njc * 350.0 + c_stage;
            else
                B = (double)njc * ((ncen * g.wseg + g.threads - 1) / g.threads) * (round_up(g.wlen, ASW_WB) + 2) * c_w +
                    (njc > 1 ? njc * 400.0 : 0.0) +               // extra barriers of the chunked form
                    (double)((g.nL * (g.Dc / 4) + g.threads - 1) / g.threads) * c_e +
                    (double)((g.nL + g.nR + g.threads - 1) / g.threads) * c_stage;
            const double d_util = (double)nD / ((double)nch * g.Dc);
            const int xt = (W + g.Tx - 1) / g.Tx;
            const double x_util = (double)W / ((double)xt * g.Tx);
            const double nwg = (double)xt * std::max(rows, 1) * nch, slots = 256.0 * k;
            const double tail = nwg / (std::ceil(nwg / slots) * slots);
            const double overlap = k > 1 ? 1.05 : 1.0;             // independent groups hide each other's build phase
            // the busiest SIMD carries k*per_simd waves: a group's time scales with per_simd, and fewer
            // resident waves hide less latency (measured: 2 waves/SIMD ~0.85x, 1 wave/SIMD ~0.6x of 3)
            const int wps = k * per_simd;
            const double occ = wps >= 3 ? 1.0 : (wps == 2 ? 0.85 : 0.6);
            const double useful = (double)win * Rx * ASW_RD * c_tap;        // = M for the 8-column tile
            const double score = (double)XG * DG / per_simd * occ * (useful / (M + B)) * d_util * x_util * tail * overlap;
            if (score > best_score) { best_score = score; best = g; found = true; }
            if (shortlist) {
                auto &slot = classes[{piped ? 80 : Rx, std::min(g.JC, 64), nch, waves}];
                if (score > slot.first) slot = {score, g};
            }
        }
        }
        if (DG <= 2) break;
    }
    // Measured exception to the cost model: with the 8-column tile and many disparity groups (DG >= 33, i.e. narrow
    // x tiles under long weight rows) staging the tap columns in chunks of 16 is 1-1.5 % FASTER than whole rows --
    // build and aggregation phases of different waves interleave (1080p/193: 46.9 -> 46.2 ms, 4K/257: 271.9 -> 269.2 ms,
    // 1080p/129: 32.4 -> 32.0 ms) -- while for DG <= 25 it is 4-6 % slower, as the model says.
    if (found && !best.pipe && best.Rx == 8 && best.JC >= win && best.DG >= 33 && win > 16) {
        AswGeom g;
        if (asw_layout(g, win, best.XG, best.DG, 160 * 1024, 16, 8)) {
            g.nchunks = best.nchunks;
            best = g;
        }
    }
    if (found && !best.pipe) asw_pick_e_scheme(best, win);      // (the phase-shifted form competed in the search above)
    // small disparity ranges: the wave kernel takes over (the workgroup geometry stays as its fallback)
    const int wave_rx = found ? asw_wave_pick(win, nD) : 0;
    if (wave_rx) best.wave_rx = wave_rx;
    if (shortlist && found) {
        std::vector<std::pair<double, AswGeom>> v;
        for (auto &kv : classes) v.push_back(kv.second);
        std::sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
        shortlist->clear();
        if (wave_rx) {          // the model's choice first, then the other tile of the wave kernel, then workgroup geometries
            shortlist->push_back(best);
            AswWaveGeom wg;
            if (!tune().wave_rx && asw_wave_layout(wg, win, nD, 12 - wave_rx)) {
                AswGeom other = best;
                other.wave_rx = 12 - wave_rx;
                shortlist->push_back(other);
            }
            if (wave_rx == 4 && asw_wave_layout(wg, win, nD, 4) && wg.RD == 6) {      // ... and the four-per-lane strip next to the six-per-lane one
                AswGeom other = best;
                other.wave_rx = 4 | 16;
                shortlist->push_back(other);
            }
        }
        // every class enters in its phase-shifted form where that exists AND in the plain form: which of the two is
        // faster depends on the tile (waves per SIMD, centres per thread), and the trials measure it
        // (next to the wave kernel only the three best workgroup classes: they have not won a trial for such ranges)
        for (size_t i = 0; i < v.size() && i < (wave_rx ? 3u : 12u) && v[i].first > 0.6 * best_score; ++i) {
            if (!v[i].second.pipe) asw_pick_e_scheme(v[i].second, win);
            v[i].second.wave_rx = 0;
            shortlist->push_back(v[i].second);
        }
    }
    return found ? SSAMD_OK : fail(SSAMD_ELIMIT, "no ASW launch geometry fits LDS for winSize=%d nD=%d", win, nD);
}

// Proximity weights exp(-|t|/gammaP) of the window taps (_passive.cpp:360-364), cached per (winSize, gammaP).
int get_prox(Ctx &c, int win, double gammaP, hipStream_t s, const float **out)
{
    if (TableEntry *e = c.proxTabs.find(win, gammaP)) { *out = (const float *)e->dev.ptr; return SSAMD_OK; }
    if (c.proxTabs.entries.size() >= c.proxTabs.max_entries) {
        // evicting frees device memory a launch in flight may still read: the one place that waits (rare: more
        // than eight parameter sets alternating on one device)
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(c.proxTabs.entries.back().dev.ptr);
        c.proxTabs.entries.pop_back();
    }
    c.proxTabs.entries.emplace_front();
    TableEntry &e = c.proxTabs.entries.front();
    e.k0 = win; e.k1 = gammaP;
    const int p = win / 2;
    e.host.resize((size_t)win * win);
    for (int i = 0; i < win; ++i)
        for (int j = 0; j < win; ++j) {
            const double di = i - p, dj = j - p;
#if SSAMD_W_FOLD
            e.host[(size_t)i * win + j] = (float)(-std::sqrt(di * di + dj * dj) / gammaP * 1.4426950408889634);      // log2 of the weight
#else
            e.host[(size_t)i * win + j] = (float)glibc_exp(-std::sqrt(di * di + dj * dj) / gammaP);      // (restated exp: see get_prox64)
#endif
        }
    int rc = e.dev.reserve(e.host.size() * 4);
    if (rc) { c.proxTabs.entries.pop_front(); return rc; }
    hipError_t he = hipMemcpyAsync(e.dev.ptr, e.host.data(), e.host.size() * 4, hipMemcpyHostToDevice, s);
    if (he != hipSuccess) {
        (void)hipFree(e.dev.ptr);
        c.proxTabs.entries.pop_front();
        return fail(SSAMD_EHIP, "hipMemcpyAsync(proximity table) failed: %s", hipGetErrorString(he));
    }
    *out = (const float *)e.dev.ptr;
    return SSAMD_OK;
}

// The same table in fp64 for the tie-break pass -- the reference's expression (_passive.cpp:360-364:
// exp(-sqrt(pow(i-padding,2) + pow(j-padding,2))/gammaP)) with glibc's exp restated on the host side (round 6: rounds 1-5 called the
// host's libm here, which made the exact mode's bit-identity a property of the host).
int get_prox64(Ctx &c, int win, double gammaP, hipStream_t s, const double **out)
{
    if (TableEntry *e = c.proxTabs64.find(win, gammaP)) { *out = (const double *)e->dev.ptr; return SSAMD_OK; }
    if (c.proxTabs64.entries.size() >= c.proxTabs64.max_entries) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(c.proxTabs64.entries.back().dev.ptr);
        c.proxTabs64.entries.pop_back();
    }
    c.proxTabs64.entries.emplace_front();
    TableEntry &e = c.proxTabs64.entries.front();
    e.k0 = win; e.k1 = gammaP;
    const int p = win / 2;
    e.host64.resize((size_t)win * win);
    for (int i = 0; i < win; ++i)
        for (int j = 0; j < win; ++j)
            {
                // exp: glibc's algorithm restated (glibc_math.hip.h), not the host's libm -- bit-identical to glibc's on every argument
                // oracle/libm_check.c tries, and the same on a host that runs another libm.  pow(int, 2) is exact, sqrt and the
                // division are IEEE operations (one correctly rounded result on any conforming host)
                const double di = i - p, dj = j - p;
                e.host64[(size_t)i * win + j] = glibc_exp(-std::sqrt(di * di + dj * dj) / gammaP);
            }
    int rc = e.dev.reserve(e.host64.size() * 8);
    if (rc) { c.proxTabs64.entries.pop_front(); return rc; }
    hipError_t he = hipMemcpyAsync(e.dev.ptr, e.host64.data(), e.host64.size() * 8, hipMemcpyHostToDevice, s);
    if (he != hipSuccess) {
        (void)hipFree(e.dev.ptr);
        c.proxTabs64.entries.pop_front();
        return fail(SSAMD_EHIP, "hipMemcpyAsync(fp64 proximity table) failed: %s", hipGetErrorString(he));
    }
    *out = (const double *)e.dev.ptr;
    return SSAMD_OK;
}

// fp64 tie-break pass (asw_exact_kernels.hip.h), part 1 -- BEFORE the aggregation: queue, pixel flags and counters of this call;
// fills the AswExactQueue the aggregation kernels append their near-ties to (round 6: no cost-image volume).
int asw_exact_prepare(Ctx &c, int W, int rows, int win, int nD, double gammaC, bool consistent, bool direct, hipStream_t s,
                      AswExactQueue &q, AswExactQueue &raw, AswExactQueue &kernel_q)
{
    const size_t nout = (size_t)rows * W;
    if (nout >= ((size_t)1 << 32)) return fail(SSAMD_ELIMIT, "exact mode: more than 2^32 output pixels per call");
    // queue: room for a few candidates of every pixel, bounded (a frame of saturated noise can flag every candidate of every
    // pixel; an overflow leaves the fp32 map and is reported: `exact_overflow`); frames of up to 4M candidates in all get room
    // for every one of them on both sides -- a flat test image cannot overflow
    // (a candidate can be queued once per side -- by a select and by a merge, or by the left and the right escalation)
    const size_t all_cands = 2 * (nout * (size_t)nD + nout);
    size_t cap = std::min<size_t>(std::max<size_t>(4 * nout, std::min<size_t>(all_cands, (size_t)1 << 23)), (size_t)1 << 26);
    if (tune().exact_cap) cap = (size_t)tune().exact_cap;
    // raw queue of merging calls (12 B per entry): what workgroups select against their tile-local winners
    size_t rawcap = direct ? 0 : std::min<size_t>(std::max<size_t>(2 * nout, std::min<size_t>(all_cands, (size_t)1 << 23)), (size_t)1 << 26);
    if (!direct && tune().exact_cap) rawcap = std::max<size_t>(rawcap, cap);
    if (!direct && tune().exact_rawcap) rawcap = (size_t)tune().exact_rawcap;
    int rc;
    // [64 B counters][flagL][flagR][zeroL][zeroR] (nout bytes each) [zrow rows]: one buffer, one memset
    if ((rc = c.xqueue.reserve(cap * 8)) || (rc = c.xcost.reserve(cap * 8)) || (rc = c.xflags.reserve(64 + 4 * nout + (size_t)rows)) ||
        (rc = c.xslots.reserve(nout * 32)) || (rawcap && (rc = c.xraw.reserve(rawcap * 12))))
        return rc;
    c.xcap = (unsigned int)cap;
    c.xrawcap = (unsigned int)rawcap;
    HIP_TRY(hipMemsetAsync(c.xflags.ptr, 0, 64 + 4 * nout + (size_t)rows, s));
    q.entries = (u64 *)c.xqueue.ptr;
    q.ekeys = nullptr;
    q.counter = (unsigned int *)c.xflags.ptr;
    q.flagL = (unsigned char *)c.xflags.ptr + 64;
    q.flagR = consistent ? q.flagL + nout : nullptr;
    q.zeroL = q.flagL + 2 * nout;
    q.zeroR = consistent ? q.flagL + 3 * nout : nullptr;
    q.zrow = q.flagL + 4 * nout;
    q.W = (unsigned int)W;
    q.cap = (unsigned int)cap;
    // Near-tie band.  128 ulps = 1.5e-5 relative at gammaC = 5 and a 35 x 35 window: the kernels' support weights inherit the
    // rounding of the Lab records to float (|dLab| <= ~1.3e-5 per colour distance), i.e. a relative error of ~2.6e-5 / gammaC on a
    // weight product -- scaled up for smaller gammaC; and the (N, S') sums are fp32 sums of win^2 products whose rounding errors
    // add up like a random walk (~win ulps): scaled up with the window beyond 35 (a fixed band ignored the tap
    // count; tests/test_gpu_exact.py::test_large_windows_against_the_oracle)
    q.tol = (uint32_t)std::min(1.0e6, (double)tune().exact_tol * std::max(1.0, 5.0 / gammaC) * std::max(1.0, win / 35.0));
    // rounding noise of the reference's fp64 quotient sum(w e) / sum(w) over n = win^2 taps: <= ~(n + 3) u relative on numerator and
    // denominator each, u = 2^-53, i.e. 2 (n + 3) u 40 absolute near the cap; two candidates can swap places when they are
    // closer than twice that (x 1.5 margin)
    q.sat_abs = (float)(1.5 * 2.0 * 2.0 * ((double)win * win + 3.0) * 1.1102230246251565e-16 * 40.0);
    q.deep = 0xffffffffu;
    raw = AswExactQueue{};
    if (!direct) {
        raw = q;
        union { float f; uint32_t u; } sa; sa.f = q.sat_abs;
        raw.deep = 0xC0000000u - sa.u;                                   // images with 40 - cost <= sat_abs stay out of the raw queue
        raw.entries = (u64 *)c.xraw.ptr;
        raw.ekeys = (uint32_t *)((u64 *)c.xraw.ptr + rawcap);
        raw.counter = q.counter + 4;
        raw.flagL = raw.flagR = nullptr;
        raw.cap = (unsigned int)rawcap;
    }
    kernel_q = direct ? q : raw;
    return SSAMD_OK;
}

// ... part 2 -- AFTER the aggregation: the queue holds the near-ties of every winner, c.keyL / c.keyR (or the map itself when the
// aggregation wrote it: `direct`) the fp32 winners.  Rewrites the winners of pixels whose near-ties fp64 decides differently.
int asw_exact_pass(Ctx &c, const AswExactQueue &q, const AswExactQueue &raw, int H, int W, int row0, int rows, int win, int maxD, int minD, double gammaC, double gammaP,
                   bool consistent, bool direct, int16_t *d_disp, hipStream_t s)
{
    const int p = win / 2;
    const size_t nout = (size_t)rows * W, npix = (size_t)H * W;
    int rc;
    const double *d_prox = nullptr;
    if ((rc = get_prox64(c, win, gammaP, s, &d_prox))) return rc;
    if ((rc = c.xlabL.reserve(npix * 24)) || (rc = c.xlabR.reserve(npix * 24))) return rc;
    AswExactArgs x;
    x.recL = (const PixRec *)c.recL.ptr; x.recR = (const PixRec *)c.recR.ptr;
    x.labL = (const double *)c.xlabL.ptr; x.labR = (const double *)c.xlabR.ptr;
    x.prox = d_prox;
    x.keyL = direct ? nullptr : (u64 *)c.keyL.ptr; x.keyR = consistent ? (u64 *)c.keyR.ptr : nullptr;
    x.disp = d_disp;
    x.q = q;
    x.raw = raw;
    x.ecost = (double *)c.xcost.ptr;
    x.costL = (u64 *)c.xslots.ptr; x.costR = x.costL + nout;
    x.idxL = (uint32_t *)(x.costR + nout); x.idxR = x.idxL + nout;
    x.wslotL = x.idxR + nout; x.wslotR = x.wslotL + nout;
    // weight tables of flagged pixels: up to 64 MB of them (6 800 pixels at a 35 x 35 window; the bench frame flags 133, the 4K frame 3 044)
    {
        const size_t per = (size_t)win * win * 8;
        size_t wcap = std::min<size_t>(65536, ((size_t)64 << 20) / per);
        if (nout >= ((size_t)1 << 31)) wcap = 0;
        if (wcap && (rc = c.xwtab.reserve(wcap * 8 + wcap * per))) return rc;
        x.wcap = (unsigned int)wcap;
        x.wpix = (uint32_t *)c.xwtab.ptr;
        x.wtab = (double *)((char *)c.xwtab.ptr + wcap * 8);
    }
    x.H = H; x.W = W; x.win = win; x.pad = p; x.minD = minD; x.maxD = maxD; x.row0 = row0; x.rows = rows;
    x.gammaC = gammaC;
    Timed t(c, s, SSAMD_K_ASW_EXACT);
    {
        const int r0 = std::max(0, row0 - p), r1 = std::min(H, row0 + rows + p);
        const long long np2 = (long long)(r1 - r0) * W;
        const int blocks = (int)std::min<long long>((2 * np2 + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(bgr2lab_f64_pair_kernel, dim3(blocks), dim3(256), 0, s, x.recL + (size_t)r0 * W, x.recR + (size_t)r0 * W,
                           (double *)c.xlabL.ptr + 3 * (size_t)r0 * W, (double *)c.xlabR.ptr + 3 * (size_t)r0 * W, np2);
    }
    const int pb = (int)std::min<long long>(((long long)nout + 255) / 256, 256 * 8);
    if (raw.entries) {
        hipLaunchKernelGGL(asw_exact_filter_kernel, dim3(256 * 8), dim3(256), 0, s, x);
        hipLaunchKernelGGL(asw_exact_escalate_kernel, dim3(pb), dim3(256), 0, s, x);
    }
    {
        const int zwin = std::min(win, EXACT_ZWIN_MAX - 1);
        const size_t zlds = (size_t)2 * zwin * (EXACT_ZSEG + 2 * (zwin / 2)) * 4;                      // two window-row tiles (<= 63 x 126 x 4 B x 2)
        if ((rc = grant_dyn_lds(c, (const void *)asw_exact_zero_kernel, (int)zlds))) return rc;
        const long long segs = (long long)rows * ((W + EXACT_ZSEG - 1) / EXACT_ZSEG);
        hipLaunchKernelGGL(asw_exact_zero_kernel, dim3((unsigned)std::min<long long>(segs, 256 * 64)), dim3(256), zlds, s, x);
    }
    hipLaunchKernelGGL(asw_exact_winners_kernel, dim3(pb), dim3(256), 0, s, x);
    if (x.wcap) hipLaunchKernelGGL(asw_exact_wtab_kernel, dim3(256 * 4), dim3(256), 0, s, x);
    hipLaunchKernelGGL(asw_exact_eval_kernel, dim3((unsigned)(c.cus * 5)), dim3(64 * EXACT_WAVES), 0, s, x);      // five 4-wave groups per CU are resident (91 VGPRs)
    hipLaunchKernelGGL(asw_exact_resolve_kernel, dim3(256 * 4), dim3(256), 0, s, x);
    hipLaunchKernelGGL(asw_exact_patch_kernel, dim3(pb), dim3(256), 0, s, x);
    HIP_TRY(hipGetLastError());
    ++c.exact_calls;
    return SSAMD_OK;
}

int launch_finalize(Ctx &c, int slot, bool lrcheck, int rows, int W, int16_t *d_disp, hipStream_t s,
                    int16_t *d_raw_right = nullptr)
{
    if (rows <= 0) return SSAMD_OK;
    Timed t(c, s, slot);
    if (d_raw_right) {          // verification dump: both raw argmins instead of the left-right check and filling
        const long long n = (long long)rows * W;
        const int blocks = (int)std::min<long long>((n + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(wta_decode_kernel, dim3(blocks), dim3(256), 0, s, (const u64 *)c.keyL.ptr, d_disp, rows, W, 0);
        hipLaunchKernelGGL(wta_decode_kernel, dim3(blocks), dim3(256), 0, s, (const u64 *)c.keyR.ptr, d_raw_right, rows, W, 1);
    } else if (lrcheck) {
        const size_t lds = (((size_t)W * 2 + 15) & ~(size_t)15) + W;      // up to 96 KiB at the 32767-column limit
        int rc = grant_dyn_lds(c, (const void *)lr_check_fill_kernel, (int)lds);
        if (rc) return rc;
        hipLaunchKernelGGL(lr_check_fill_kernel, dim3(rows), dim3(256), lds, s, (const u64 *)c.keyL.ptr,
                           (const u64 *)c.keyR.ptr, d_disp, rows, W);
    } else {
        const long long n = (long long)rows * W;
        const int blocks = (int)std::min<long long>((n + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(wta_decode_kernel, dim3(blocks), dim3(256), 0, s, (const u64 *)c.keyL.ptr, d_disp, rows, W, 0);
    }
    HIP_TRY(hipGetLastError());
    return SSAMD_OK;
}

// rm != nullptr: dL / dR are unused; the pixel records come from the RAW frames through the rig's maps
// (remap_lab_records_pair_kernel: rectification + Lab in one launch)
int asw_device_impl(Ctx &c, const uint8_t *dL, const uint8_t *dR, int H, int W, int row0, int rows, int win,
                    int maxD, int minD, double gammaC, double gammaP, int consistent, int16_t *d_disp,
                    float *d_costs, hipStream_t s, bool alternate = false, int16_t *d_raw_right = nullptr, const RemapSrc *rm = nullptr,
                    bool exact = false, int skip_at = 0, int skip = 0)
{
    // skip > 0: rows [row0 + skip_at, row0 + skip_at + skip) of the range are NOT matched (ssamd_asw_device_rows2: the two border
    // bands of a row strip in one launch); buffers stay laid out for the whole range [row0, row0 + rows)
    int rc = check_common(H, W, win, minD, maxD, row0, rows);
    if (rc) return rc;
    if (skip < 0 || skip_at < 0 || skip_at + skip > rows) return fail(SSAMD_EINVAL, "bad row gap [%d,%d) in a range of %d rows", skip_at, skip_at + skip, rows);
    if (skip > 0 && (alternate || d_costs || d_raw_right)) return fail(SSAMD_EINVAL, "two row ranges: plain, consistent and exact matching only");
    if (skip == rows) return SSAMD_OK;
    if (!(gammaC > 0) || !(gammaP > 0)) return fail(SSAMD_EINVAL, "gammaC and gammaP must be positive");
    if (exact && (alternate || d_costs)) return fail(SSAMD_EINVAL, "the exact (fp64 tie-break) mode has no alternate-rows form and no cost dump");
    if (maxD < minD) exact = false;                                   // empty candidate loops: nothing to break ties between
    // alternate-rows mode: row0 is matched exactly, then every second row; the range must end with an exact row or with
    // the image (asw_alternate_rows arranges that for strips)
    if (alternate && d_costs) return fail(SSAMD_EINVAL, "the alternate-rows mode has no cost dump");
    if (rows == 0) return SSAMD_OK;
    ScratchOrder order(c, s);
    const int p = win / 2, nD = maxD - minD + 1;
    const size_t npix = (size_t)H * W, nout = (size_t)rows * W;

    // One disparity chunk and no right-referenced pass: each pixel is decided by exactly one workgroup, which then
    // writes the disparity itself -- no key buffer, atomics or decode kernel (34 instead of 48+ bytes of HBM per pixel).
    AswArgs a{};
    AswExactQueue xq_final{}, xq_raw{};                               // exact mode: the final near-tie queue and, for merging calls, the raw one
    const int grows = alternate ? (rows + 1) / 2 : rows - skip;       // workgroup rows: every row (of both ranges), or the even ones
    if (nD >= 1 && (rc = asw_choose_geometry(a.g, W, grows, win, nD))) return rc;
    // Autotuning (ssamd_autotune): the first call for a problem shape times the best geometry of every class of
    // candidates on the real buffers and keeps the fastest.  Every geometry accumulates the same taps in the same
    // order, so the result does not depend on the choice (and the trial launches are idempotent).
    const std::array<int, 4> shape{W, grows, win, nD};
    std::vector<AswGeom> trial;
    const double call_taps = (double)W * grows * nD * win * win;
    const int tune_mode = g_autotune.load();
    const bool tune_now = (tune_mode > 0 || (tune_mode < 0 && call_taps <= ASW_AUTOTUNE_SMALL_TAPS));
    bool tuned_already;
    { std::lock_guard<std::mutex> glk(g_geom_mutex); tuned_already = g_asw_geom_tuned.count(shape) != 0; }
    if (tune_now && nD >= 1 && !asw_geometry_forced() && !tuned_already) {
        AswGeom tmp;
        if (asw_search_geometry(tmp, W, grows, win, nD, &trial) != SSAMD_OK || trial.size() < 2) trial.clear();
    }
    // The TAD volume is scratch of THIS library next to the caller's own allocations (torch's caching allocator on the
    // same GPU): never more than 24 GiB and never more than half of what is free right now (plus what the buffer already
    // holds).  The phase-shifted kernel builds its e tiles itself when there is no volume; the wave kernel cannot, so
    // a range it would serve falls back to the workgroup geometry stored next to it.
    // Free memory is only asked for (a driver query, tens of microseconds next to a 0.1 ms Tsukuba call) when a volume
    // would have to GROW: what fits the buffer the context already holds fits.
    const size_t evol_cap_max = tune().evol_max_mb ? std::min((size_t)24 << 30, (size_t)tune().evol_max_mb << 20) : (size_t)24 << 30;
    size_t evol_limit_cached = 0;
    bool evol_limit_known = false;
    auto evol_fits = [&](size_t bytes) {
        if (bytes > evol_cap_max) return false;
        if (bytes <= c.evol.cap) return true;
        if (!evol_limit_known) {
            size_t free_b = 0, total_b = 0;
            evol_limit_cached = evol_cap_max;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) evol_limit_cached = std::min(evol_cap_max, c.evol.cap + free_b / 2);
            else (void)hipGetLastError();
            evol_limit_known = true;
        }
        return bytes <= evol_limit_cached;
    };
    auto wave_volume_fits = [&](const AswGeom &g) {
        AswWaveGeom wg;
        if (!g.wave_rx || !asw_wave_layout(wg, win, nD, g.wave_rx)) return !g.wave_rx;
        const int xt = (W + wg.Txw - 1) / wg.Txw, erows = std::min(H, row0 + rows + win / 2) - std::max(0, row0 - win / 2);
        return evol_fits((size_t)erows * (size_t)round_up(xt * wg.Txw + 2 * (win / 2), 4) * (size_t)wg.Se + 4096);
    };
    if (nD >= 1 && a.g.wave_rx && !wave_volume_fits(a.g)) ++c.evol_fallbacks;
    if (nD >= 1 && !wave_volume_fits(a.g)) a.g.wave_rx = 0;
    trial.erase(std::remove_if(trial.begin(), trial.end(), [&](const AswGeom &g) { return !wave_volume_fits(g); }), trial.end());
    if (trial.size() < 2) trial.clear();
    auto is_direct = [&](const AswGeom &g) { return nD >= 1 && (g.nchunks == 1 || g.wave_rx) && !consistent; };
    bool need_keys = !is_direct(a.g) || alternate;  // the alternate mode merges its odd-row jobs through the left keys
    for (const AswGeom &g : trial) need_keys = need_keys || !is_direct(g);
    if (need_keys) {
        if ((rc = c.keyL.reserve(nout * 8))) return rc;
        HIP_TRY(hipMemsetAsync(c.keyL.ptr, 0xFF, nout * 8, s));
    }
    if (consistent) {
        if ((rc = c.keyR.reserve(nout * 8))) return rc;
        HIP_TRY(hipMemsetAsync(c.keyR.ptr, 0xFF, nout * 8, s));
    }

    if (nD >= 1) {
        if ((rc = c.recL.reserve(npix * sizeof(PixRec)))) return rc;
        if ((rc = c.recR.reserve(npix * sizeof(PixRec)))) return rc;
        const float *d_prox = nullptr;
        if ((rc = get_prox(c, win, gammaP, s, &d_prox))) return rc;
        const int r0 = std::max(0, row0 - p), r1 = std::min(H, row0 + rows + p);
        const long long np2 = (long long)(r1 - r0) * W;
        const int lab_blocks = (int)std::min<long long>((2 * np2 + 255) / 256, 256 * 8);
        auto launch_lab = [&]() -> int {       // Lab records of both images, one launch
            Timed t(c, s, SSAMD_K_LAB);
            if (rm)
                hipLaunchKernelGGL(remap_lab_records_pair_kernel, dim3(lab_blocks), dim3(256), 0, s, *rm, (PixRec *)c.recL.ptr, (PixRec *)c.recR.ptr,
                                   (long long)r0 * W, np2);
            else
                hipLaunchKernelGGL(bgr2lab_records_pair_kernel, dim3(lab_blocks), dim3(256), 0, s, dL + (size_t)r0 * W * 3, dR + (size_t)r0 * W * 3,
                                   (PixRec *)c.recL.ptr + (size_t)r0 * W, (PixRec *)c.recR.ptr + (size_t)r0 * W, np2);
            HIP_TRY(hipGetLastError());
            return SSAMD_OK;
        };
        // Round 5: when the call goes straight to its final geometry (no trial launches) and the images are plain byte arrays, the
        // records are NOT launched here: the TAD volume is then built from the images' bytes, independent of the records, and both
        // jobs share one launch (asw_prepass_kernel, see prepare_evol) -- two dependent launches per call instead of three.
        bool lab_pending = trial.empty() && !rm && tune().prepass_fuse != 0;
        if (!lab_pending && (rc = launch_lab())) return rc;

        a.recL = (const PixRec *)c.recL.ptr; a.recR = (const PixRec *)c.recR.ptr;
        a.prox = d_prox;
        a.keyR = consistent ? (u64 *)c.keyR.ptr : nullptr;
        a.costs = d_costs;
        a.cost_keys = 0;
        a.xq = AswExactQueue{};                                       // (the autotuner's trial launches run without the queue)
        a.H = H; a.W = W; a.win = win; a.pad = p; a.minD = minD; a.maxD = maxD; a.row0 = row0; a.rows = rows;
        a.kC = (float)(-1.4426950408889634 / gammaC);
        a.ystep = alternate ? 2 : 1;
        a.yskip_at = skip > 0 ? skip_at : 0x7fffffff; a.yskip = skip;
        a.evol = nullptr; a.erow0 = r0; a.erows = r1 - r0; a.evolW = 0;
        // pre-computed truncated-absolute-difference volume for the phase-shifted kernel (asw_tad_volume_kernel);
        // SSAMD_ASW_EVOL=0 keeps the in-kernel e tiles (experiments / tests)
        AswWaveArgs wa;
        auto prepare_evol = [&](const AswGeom &g) -> int {
            a.evol = nullptr;
            int chunks = g.nchunks, Tx = g.Tx, Dc = g.Dc, Se = g.Se;
            if (g.wave_rx) {
                if (!asw_wave_layout(wa.g, win, nD, g.wave_rx, !d_costs && tune().wave_unroll != 0))
                    return fail(SSAMD_ELIMIT, "wave kernel geometry does not fit LDS");
                chunks = 1; Tx = wa.g.Txw; Dc = wa.g.Dc; Se = wa.g.Se;
            } else if (!g.pipe || tune().asw_evol == 0) {
                return SSAMD_OK;
            }
            const int xt = (W + Tx - 1) / Tx;
            // rows stay 16-byte aligned for any Se; the phase-shifted kernel's half-width tail tiles (see launch) may reach
            // up to half a tile + 4 columns further than the last full tile
            const int evolW = round_up(xt * Tx + 2 * p + (g.wave_rx ? 0 : Tx / 2 + 8), 4);
            const size_t bytes = (size_t)chunks * (size_t)(r1 - r0) * (size_t)evolW * (size_t)Se;
            // A buffer four times larger than the calls need is given back -- but only after eight such calls in a row
            // and never between the trial launches of the autotuner: a workload alternating a large and a small shape
            // (or the tuner's round-robin over wave and workgroup candidates) must not pay a device synchronisation,
            // a hipFree and a hipMalloc per switch.
            if (c.evol.cap > ((size_t)256 << 20) && (bytes + 4096) * 4 < c.evol.cap) {
                if (trial.empty() && ++c.evol_small_calls >= 8) {
                    (void)hipStreamSynchronize(s);                    // (earlier launches of this call may still read it)
                    c.evol.release();
                    c.evol_small_calls = 0;
                }
            } else {
                c.evol_small_calls = 0;
            }
            int erc = SSAMD_ENOMEM;
            if (tune().evol_fail) {                                   // test hook: a hipMalloc that really fails
                void *none = nullptr;
                size_t free_b = 0, total_b = (size_t)512 << 30;
                (void)hipMemGetInfo(&free_b, &total_b);
                if (hipMalloc(&none, 2 * total_b) == hipSuccess) (void)hipFree(none);     // twice the device's memory
                // (deliberately NOT drained here: the fallback below has to cope with the sticky error itself)
            } else if (evol_fits(bytes + 4096)) {
                erc = c.evol.reserve(bytes + 4096);                   // + one DMA piece of slack behind the last tile
            }
            if (erc) {
                (void)hipGetLastError();      // a failed hipMalloc stays the thread's last HIP error on ROCm 7: drop it before the launches' checks
                // the phase-shifted kernel builds its e tiles itself when there is no volume (A.evol == nullptr);
                // only the wave kernel cannot run without one
                if (g.wave_rx) return fail(erc == SSAMD_ENOMEM ? SSAMD_ENOMEM : erc, "TAD volume of %zu bytes does not fit the device memory left", bytes);
                ++c.evol_fallbacks;
                g_err.clear();
                return SSAMD_OK;
            }
            a.evol = (const unsigned char *)c.evol.ptr;
            a.evolW = evolW;
            Timed t(c, s, SSAMD_K_LAB);
            const dim3 egrid((unsigned)((evolW + TADV_COLS - 1) / TADV_COLS), (unsigned)(r1 - r0), (unsigned)chunks);
            const size_t elds = (size_t)(2 * TADV_COLS + Dc) * 4;
            const long long tiles = (long long)egrid.x * egrid.y * egrid.z;
            if (lab_pending && tiles + lab_blocks < (1ll << 31)) {
                AswPrepassArgs pa;
                pa.bgrL = dL; pa.bgrR = dR; pa.recL = (PixRec *)c.recL.ptr; pa.recR = (PixRec *)c.recR.ptr;
                pa.evol = (unsigned char *)c.evol.ptr; pa.npix_total = (long long)H * W;
                pa.W = W; pa.pad = p; pa.minD = minD; pa.Dc = Dc; pa.Se = Se; pa.erow0 = r0; pa.erows = r1 - r0; pa.evolW = evolW;
                pa.rd = g.wave_rx ? wa.g.RD : 4;
                pa.lab_blocks = lab_blocks; pa.ex = (int)egrid.x; pa.ey = (int)egrid.y;
                if (int grc = grant_dyn_lds(c, (const void *)asw_prepass_kernel, (int)elds)) return grc;
                hipLaunchKernelGGL(asw_prepass_kernel, dim3((unsigned)(tiles + lab_blocks)), dim3(256), elds, s, pa);
                HIP_TRY(hipGetLastError());
                lab_pending = false;
                return SSAMD_OK;
            }
            if (lab_pending) {                   // (cannot share a launch: records first, as in rounds 2-4)
                lab_pending = false;
                if (int lrc = launch_lab()) return lrc;
            }
            if (int grc = grant_dyn_lds(c, (const void *)asw_tad_volume_kernel, (int)elds)) return grc;
            hipLaunchKernelGGL(asw_tad_volume_kernel, egrid, dim3(256), elds, s, (const PixRec *)c.recL.ptr, (const PixRec *)c.recR.ptr,
                               (unsigned char *)c.evol.ptr, W, p, minD, Dc, Se, r0, r1 - r0, evolW, g.wave_rx ? wa.g.RD : 4);
            HIP_TRY(hipGetLastError());
            return SSAMD_OK;
        };
        auto launch = [&](const AswGeom &g) -> int {
            a.g = g;
            a.keyL = is_direct(g) ? nullptr : (u64 *)c.keyL.ptr;
            a.disp = is_direct(g) ? d_disp : nullptr;
            if (g.wave_rx) {                  // (prepare_evol(g) filled wa.g and built the volume)
                wa.recL = a.recL; wa.recR = a.recR; wa.prox = a.prox;
                wa.keyL = a.keyL; wa.keyR = a.keyR; wa.disp = a.disp; wa.costs = a.costs; wa.cost_keys = a.cost_keys; wa.xq = a.xq;
                wa.evol = a.evol; wa.erow0 = a.erow0; wa.erows = a.erows; wa.evolW = a.evolW;
                wa.H = H; wa.W = W; wa.win = win; wa.pad = p; wa.minD = minD; wa.maxD = maxD; wa.row0 = row0; wa.rows = rows;
                wa.ystep = a.ystep; wa.kC = a.kC; wa.yb0 = 0; wa.yskip_at = a.yskip_at; wa.yskip = a.yskip;
                auto wk = wa.g.RX == 8 ? (d_costs ? asw_aggregate_wave_kernel<true, 8> : asw_aggregate_wave_kernel<false, 8>)
                                       : (d_costs ? asw_aggregate_wave_kernel<true, 4> : asw_aggregate_wave_kernel<false, 4>);
                // build rounds known at compile time (straight-line build): the common combinations
                const int kl = (wa.g.Txw + 63) / 64, kr = (wa.g.nRcw + 63) / 64;
                const bool unrolled = tune().wave_unroll != 0;
                if (wa.g.RD == 6) {                                                         // six disparities per lane
                    wk = d_costs ? asw_aggregate_wave6_kernel<true, 0> : asw_aggregate_wave6_kernel<false, 0>;
                    if (unrolled && !d_costs && wa.g.K == 3) wk = wa.g.creg ? asw_aggregate_wave6_kernel<false, 3, true> : asw_aggregate_wave6_kernel<false, 3>;
                    else if (unrolled && !d_costs && wa.g.K == 2) wk = wa.g.creg ? asw_aggregate_wave6_kernel<false, 2, true> : asw_aggregate_wave6_kernel<false, 2>;
                } else if (unrolled && !d_costs && wa.g.merged) {
                    const int key = wa.g.RX * 10 + wa.g.K;                                  // merged build, K rounds
                    if (key == 42) wk = wa.g.creg ? asw_aggregate_wave_kernel<false, 4, 0, 0, 2, true> : asw_aggregate_wave_kernel<false, 4, 0, 0, 2>;   // class default D 0..16 four per lane: 48 + 67 centres
                    else if (key == 43) wk = wa.g.creg ? asw_aggregate_wave_kernel<false, 4, 0, 0, 3, true> : asw_aggregate_wave_kernel<false, 4, 0, 0, 3>;
                    else if (key == 44) wk = wa.g.creg ? asw_aggregate_wave_kernel<false, 4, 0, 0, 4, true> : asw_aggregate_wave_kernel<false, 4, 0, 0, 4>;
                    else if (key == 82) wk = asw_aggregate_wave_kernel<false, 8, 0, 0, 2>;
                    else if (key == 83) wk = asw_aggregate_wave_kernel<false, 8, 0, 0, 3>;
                    else if (key == 84) wk = asw_aggregate_wave_kernel<false, 8, 0, 0, 4>;
                } else if (unrolled && !d_costs) {
                    const int key = wa.g.RX * 100 + kl * 10 + kr;
                    if (key == 822) wk = asw_aggregate_wave_kernel<false, 8, 2, 2>;         // 17..28 disparities (class default)
                    else if (key == 812) wk = asw_aggregate_wave_kernel<false, 8, 1, 2>;    // 29..48
                    else if (key == 412) wk = asw_aggregate_wave_kernel<false, 4, 1, 2>;    // 13..20
                    else if (key == 422) wk = asw_aggregate_wave_kernel<false, 4, 2, 2>;    // 9..12
                    else if (key == 423) wk = asw_aggregate_wave_kernel<false, 4, 2, 3>;    // 5..8
                }
                const int lds = wa.g.wave_lds * wa.g.waves, xt = (W + wa.g.Txw - 1) / wa.g.Txw;
                if (int grc = grant_dyn_lds(c, (const void *)wk, lds)) return grc;
                hipLaunchKernelGGL(wk, dim3((xt + wa.g.waves - 1) / wa.g.waves, grows, 1), dim3(64 * wa.g.waves), lds, s, wa);
                HIP_TRY(hipGetLastError());
                return SSAMD_OK;
            }
            const dim3 grid((W + g.Tx - 1) / g.Tx, grows, g.nchunks), block(g.threads);
            const bool chunked = g.JC < win;
            if (g.pipe) {
                auto pk = d_costs ? asw_aggregate_pipe_kernel<true> : asw_aggregate_pipe_kernel<false>;
                // strides known at compile time for the tiles of the headline configurations (immediate offsets in the
                // tap steps): 120 x 196 (1080p / D 0..192) and 88 x 260 (4096 x 2160 / D 0..256), 216 x 68 (D 0..64)
                // (with the cost / cost-image dump too: exact=True on the headline tiles ran the run-time-stride form, + 0.9 ms at 1080p / 193)
                // SSAMD_ASW_STATIC=2 (round 6 experiment): instantiations that take the WHOLE tile geometry and the window from compile-time
                // constants -- chosen only when the planned geometry equals the constexpr restatement field by field
                if (tune().asw_static != 0) {
                    if (g.SL == 120 && g.SR == 316 && g.Se == 208) pk = d_costs ? asw_aggregate_pipe_kernel<true, 120, 316, 208> : asw_aggregate_pipe_kernel<false, 120, 316, 208>;
                    else if (g.SL == 88 && g.SR == 348 && g.Se == 272) pk = d_costs ? asw_aggregate_pipe_kernel<true, 88, 348, 272> : asw_aggregate_pipe_kernel<false, 88, 348, 272>;
                    else if (g.SL == 216 && g.SR == 284 && g.Se == 80) pk = d_costs ? asw_aggregate_pipe_kernel<true, 216, 284, 80> : asw_aggregate_pipe_kernel<false, 216, 284, 80>;
                }
                if (tune().asw_static == 2 && !d_costs) {
                    auto is_tile = [&](AswPipeTileId id) { return win == id.win && asw_pipe_geom_matches(g, asw_pipe_geom_constexpr(id)); };
                    if (is_tile(AswPipeTile<120, 316, 208>::id)) pk = asw_aggregate_pipe_kernel<false, 120, 316, 208, true>;
                    else if (is_tile(AswPipeTile<88, 348, 272>::id)) pk = asw_aggregate_pipe_kernel<false, 88, 348, 272, true>;
                    else if (is_tile(AswPipeTile<216, 284, 80>::id)) pk = asw_aggregate_pipe_kernel<false, 216, 284, 80, true>;
                    else ++c.static_tile_mismatch;       // (counted: ssamd_counter "static_tile_mismatch")
                }
                const int pipe_lds = a.evol ? g.lds_bytes_evol : g.lds_bytes;          // (no staged colour bytes when the e tiles come from the volume)
                if (pipe_lds > 160 * 1024) return fail(SSAMD_ELIMIT, "this tile needs the TAD volume (LDS %d bytes without it)", pipe_lds);
                if (int grc = grant_dyn_lds(c, (const void *)pk, pipe_lds)) return grc;
                // The last PARTIAL round of workgroups (round 4).  The kernel keeps one 12-wave workgroup per CU, so a launch
                // of n workgroups takes ceil(n / 256) rounds: a row strip of an 8-GPU run (135 rows x 16 tiles = 8.44 rounds)
                // pays nine.  When the last round is at most half full, the rows that fill whole rounds keep the tile and the
                // remaining rows are launched with tiles of half the columns (twice the workgroups, half the taps each: the
                // round ends after about half its time).  Same taps in the same order per (x, d): maps cannot change
                // (tests/test_gpu_asw.py).  Worth ~4 % at 8.44 rounds, nothing beyond a few dozen; SSAMD_ASW_TAIL=0 / 1 forces.
                int rows_main = grows;
                AswGeom tail_g;
                if (!alternate && tune().asw_tail != 0 && g.XG >= 4) {
                    // workgroups in flight at a time: the device's CUs x the tile's residency (168 VGPRs -> three waves per SIMD;
                    // the tile's LDS).  One per CU for the 9- to 12-wave tiles of the headline configurations.
                    const int per_simd = (g.threads / 64 + 3) / 4;
                    const long long resident = std::max(1, std::min(3 / std::max(1, per_simd), (160 * 1024) / std::max(1, pipe_lds)));
                    const long long per_row = (long long)grid.x * g.nchunks, n = per_row * grows, slots = (long long)c.cus * resident;
                    const long long full = n / slots, rem = n - full * slots;
                    if (full >= 1 && rem > 0 && 2 * rem <= slots && (full < 32 || tune().asw_tail > 0)) {
                        const int rm = (int)(full * slots / per_row);
                        if (rm >= 1 && rm < grows &&
                            asw_layout_e(tail_g, win, (g.XG + 1) / 2, g.DG, 160 * 1024, g.JC, 8, true, false, true) && tail_g.pipe &&
                            tail_g.Se == g.Se && tail_g.Dc == g.Dc &&
                            (a.evol == nullptr || round_up(((W + tail_g.Tx - 1) / tail_g.Tx) * tail_g.Tx + 2 * p, 4) <= a.evolW)) {
                            tail_g.nchunks = g.nchunks;
                            rows_main = rm;
                        }
                    }
                }
                hipLaunchKernelGGL(pk, dim3(grid.x, rows_main, grid.z), block, pipe_lds, s, a);
                HIP_TRY(hipGetLastError());
                if (rows_main < grows) {
                    ++c.tail_splits;
                    AswArgs t = a;
                    t.g = tail_g;
                    t.yb0 = rows_main;              // (workgroup rows continue; outputs are addressed by image row, so the buffers stay put)
                    auto tk = d_costs ? asw_aggregate_pipe_kernel<true> : asw_aggregate_pipe_kernel<false>;
                    const int tail_lds = a.evol ? tail_g.lds_bytes_evol : tail_g.lds_bytes;
                    if (int grc = grant_dyn_lds(c, (const void *)tk, tail_lds)) return grc;
                    hipLaunchKernelGGL(tk, dim3((W + tail_g.Tx - 1) / tail_g.Tx, grows - rows_main, tail_g.nchunks), dim3(tail_g.threads),
                                       tail_lds, s, t);
                    HIP_TRY(hipGetLastError());
                }
                return SSAMD_OK;
            }
            auto kern = chunked ? (d_costs ? asw_aggregate_kernel<true, true> : asw_aggregate_kernel<false, true>)
                                : (d_costs ? asw_aggregate_kernel<true, false> : asw_aggregate_kernel<false, false>);
            if (g.Rx == 4)
                kern = chunked ? (d_costs ? asw_aggregate_kernel<true, true, 4> : asw_aggregate_kernel<false, true, 4>)
                               : (d_costs ? asw_aggregate_kernel<true, false, 4> : asw_aggregate_kernel<false, false, 4>);
            if (int grc = grant_dyn_lds(c, (const void *)kern, g.lds_bytes)) return grc;
            hipLaunchKernelGGL(kern, grid, block, g.lds_bytes, s, a);
            HIP_TRY(hipGetLastError());
            return SSAMD_OK;
        };
        if (!trial.empty()) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            // round-robin over the candidates, several rounds, fastest launch of each: clocks ramp up during the
            // first launches after an idle period, so timing the candidates one after the other would favour the
            // late ones
            std::vector<float> cand_ms(trial.size(), 3.0e38f);
            bool all_prepared = true;     // a candidate whose volume cannot be had right now is never launched, and the verdict of such a round is not cached
            // (a phase-shifted tile whose volume cannot be had is timed with in-kernel e tiles, or fails its launch when it was
            //  planned on the volume's LDS saving: that round says nothing about the tile, so its verdict is not kept either)
            const bool want_evol = tune().asw_evol != 0;
            for (const AswGeom &g : trial) {                                             // code load, clocks, scratch
                if (prepare_evol(g) != SSAMD_OK) { all_prepared = false; continue; }
                if (g.pipe && !g.wave_rx && want_evol && !a.evol) all_prepared = false;
                if (launch(g) != SSAMD_OK) all_prepared = false;
            }
            for (int round = 0; round < 4; ++round)
                for (size_t ci = 0; ci < trial.size(); ++ci) {
                    float ms = 3.0e38f;
                    if (hipEventRecord(e0, s) == hipSuccess && prepare_evol(trial[ci]) == SSAMD_OK && launch(trial[ci]) == SSAMD_OK &&
                        hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess)
                        (void)hipEventElapsedTime(&ms, e0, e1);
                    else
                        all_prepared = false;
                    if (trial[ci].pipe && !trial[ci].wave_rx && want_evol && !a.evol) all_prepared = false;
                    if (round > 0) cand_ms[ci] = std::min(cand_ms[ci], ms);  // round 0 is warm-up
                }
            AswGeom fastest = a.g;
            float best_ms = 3.0e38f;
            for (size_t ci = 0; ci < trial.size(); ++ci)
                // the model's own choice (first) keeps the job unless another candidate is clearly faster
                if (cand_ms[ci] < best_ms * (ci == 0 ? 1.0f : 0.985f)) { best_ms = cand_ms[ci]; fastest = trial[ci]; }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (all_prepared) {
                std::lock_guard<std::mutex> glk(g_geom_mutex);
                g_asw_geom_cache[shape] = fastest;
                g_asw_geom_tuned[shape] = true;
            } else {
                g_err.clear();
            }
            a.g = fastest;
        }
        {
            AswGeom final_geom = a.g;
            rc = prepare_evol(final_geom);
            if (rc == SSAMD_ENOMEM && final_geom.wave_rx) {
                // the volume's hipMalloc failed although the pre-check said it fits (fragmentation, another allocator on the
                // same GPU): the wave kernel cannot run without it, the workgroup geometry stored next to it can (in-kernel
                // e tiles, or a smaller volume if that one can be had)
                ++c.evol_fallbacks;
                g_err.clear();
                final_geom.wave_rx = 0;
                a.g = final_geom;
                if (!is_direct(final_geom) && !need_keys) {
                    if ((rc = c.keyL.reserve(nout * 8))) return rc;
                    HIP_TRY(hipMemsetAsync(c.keyL.ptr, 0xFF, nout * 8, s));
                    need_keys = true;
                }
                rc = prepare_evol(final_geom);
            }
            if (rc) return rc;
            if (final_geom.pipe && !final_geom.wave_rx && !a.evol && final_geom.lds_bytes > 160 * 1024) {
                // the tile was planned on the TAD volume (no staged colour bytes in LDS) and the volume cannot be had: plan
                // again for the in-kernel e tiles (asw_layout_e: t_pipe_full_lds)
                t_pipe_full_lds = true;
                AswGeom g2;
                const int r2 = asw_search_geometry(g2, W, grows, win, nD);
                t_pipe_full_lds = false;
                if (r2) return r2;
                g2.wave_rx = 0;
                if (!is_direct(g2) && !need_keys) {
                    if ((rc = c.keyL.reserve(nout * 8))) return rc;
                    HIP_TRY(hipMemsetAsync(c.keyL.ptr, 0xFF, nout * 8, s));
                }
                final_geom = g2;
                if ((rc = prepare_evol(final_geom))) return rc;
            }
            if (lab_pending) {                   // no volume for this call (in-kernel e tiles, round-1 kernel): the records on their own
                lab_pending = false;
                if ((rc = launch_lab())) return rc;
            }
            if (exact) {
                if (!trial.empty()) {
                    // the trial launches of the autotuner merged their winners into the keys: start the final launch from clean
                    // ones, or its tile-local winners would meet their own copies (asw_exact_merge)
                    if (need_keys) HIP_TRY(hipMemsetAsync(c.keyL.ptr, 0xFF, nout * 8, s));
                    if (consistent) HIP_TRY(hipMemsetAsync(c.keyR.ptr, 0xFF, nout * 8, s));
                }
                if ((rc = asw_exact_prepare(c, W, rows, win, nD, gammaC, consistent != 0, is_direct(final_geom), s, xq_final, xq_raw, a.xq))) return rc;
            }
            Timed t(c, s, SSAMD_K_ASW_AGG);
            if ((rc = launch(final_geom))) return rc;
        }
    }
    const bool direct = is_direct(a.g);
    if (exact && nD >= 1 &&
        (rc = asw_exact_pass(c, xq_final, xq_raw, H, W, row0, rows, win, maxD, minD, gammaC, gammaP, consistent != 0, direct, d_disp, s)))
        return rc;
    if (!direct && skip > 0) {
        // decode / left-right check of the two bands only (row-local: _passive.cpp:251-285); the rows between keep what the
        // interior call wrote
        const auto band = [&](int b0, int nb) -> int {
            if (nb <= 0) return SSAMD_OK;
            const size_t off = (size_t)b0 * W;
            Timed t(c, s, SSAMD_K_ASW_FIN);
            if (consistent) {
                const size_t lds = (((size_t)W * 2 + 15) & ~(size_t)15) + W;
                if (int grc = grant_dyn_lds(c, (const void *)lr_check_fill_kernel, (int)lds)) return grc;
                hipLaunchKernelGGL(lr_check_fill_kernel, dim3(nb), dim3(256), lds, s, (const u64 *)c.keyL.ptr + off, (const u64 *)c.keyR.ptr + off,
                                   d_disp + off, nb, W);
            } else {
                const long long n = (long long)nb * W;
                hipLaunchKernelGGL(wta_decode_kernel, dim3((int)std::min<long long>((n + 255) / 256, 256 * 8)), dim3(256), 0, s,
                                   (const u64 *)c.keyL.ptr + off, d_disp + off, nb, W, 0);
            }
            HIP_TRY(hipGetLastError());
            return SSAMD_OK;
        };
        if ((rc = band(0, skip_at)) || (rc = band(skip_at + skip, rows - skip_at - skip))) return rc;
    } else if (!direct && (rc = launch_finalize(c, SSAMD_K_ASW_FIN, consistent != 0, rows, W, d_disp, s, d_raw_right))) return rc;
    if (alternate && rows > 1) {
        // odd rows: candidates bounded by the exact rows above and below (asw_alt_kernels.hip.h).  With an
        // empty disparity range the decode already wrote x everywhere and the fill reproduces it.
        AswAltArgs f;
        const size_t nodd = (size_t)(rows / 2) * W;
        if ((double)nodd * ((nD + 7) / 8 + 1) >= 4.0e9)
            return fail(SSAMD_ELIMIT, "alternate-rows mode: image x disparity range too large for the 32-bit job counter");
        f.cap = (unsigned int)std::min<size_t>(std::max<size_t>(nodd, 1 << 16), 1u << 28);   // jobs of 8 candidates
        if (tune().alt_queue_cap)                                   // test hook: a tiny queue forces the in-place path
            f.cap = (unsigned int)tune().alt_queue_cap;
        if ((rc = c.altq.reserve((size_t)f.cap * 8 + 16))) return rc;
        f.ctr = (unsigned int *)c.altq.ptr; f.queue = (u64 *)((char *)c.altq.ptr + 16);
        HIP_TRY(hipMemsetAsync(f.ctr, 0, 16, s));
        f.recL = (const PixRec *)c.recL.ptr; f.recR = (const PixRec *)c.recR.ptr; f.prox = a.prox;
        f.disp = d_disp; f.key = (u64 *)c.keyL.ptr;
        f.H = H; f.W = W; f.win = win; f.pad = p; f.minD = minD; f.maxD = maxD; f.row0 = row0; f.rows = rows;
        f.kC = (float)(-1.4426950408889634 / gammaC);
        const dim3 pix_grid((W + 255) / 256, rows / 2);
        Timed t(c, s, SSAMD_K_ASW_ALT);
        hipLaunchKernelGGL(asw_alt_scan_kernel, pix_grid, dim3(256), 0, s, f);
        hipLaunchKernelGGL(asw_alt_jobs_kernel, dim3(256 * 8), dim3(256), 0, s, f);
        hipLaunchKernelGGL(asw_alt_decode_kernel, pix_grid, dim3(256), 0, s, f);
        HIP_TRY(hipGetLastError());
    }
    return SSAMD_OK;
}


// The alternate-rows mode on a row range of a (sub-)image.  row_parity = parity of the sub-image's row 0 in the whole
// image: rows whose index in the whole image is even are matched exactly, the odd ones are filled from their two exact
// neighbours -- so a range that starts or ends with an odd row also needs the exact row just outside it (the caller's
// halo is winSize/2 + 1 rows then).  Those rows are computed into a scratch map and the requested rows copied out.
int asw_alternate_rows(Ctx &c, const uint8_t *dL, const uint8_t *dR, int H, int W, int row0, int rows, int row_parity, int win,
                       int maxD, int minD, double gammaC, double gammaP, int consistent, int16_t *d_disp, hipStream_t s)
{
    int rc = check_common(H, W, win, minD, maxD, row0, rows);
    if (rc) return rc;
    if (rows == 0) return SSAMD_OK;
    const bool top_odd = ((row0 + row_parity) & 1) != 0, bottom_odd = ((row0 + rows - 1 + row_parity) & 1) != 0;
    if (top_odd && row0 == 0)
        return fail(SSAMD_EINVAL, "alternate rows: the first output row is an odd row of the image, the sub-image must start at least one row above it");
    const int e0 = top_odd ? row0 - 1 : row0, e1 = bottom_odd ? std::min(H, row0 + rows + 1) : row0 + rows;
    if (e0 == row0 && e1 == row0 + rows)
        return asw_device_impl(c, dL, dR, H, W, row0, rows, win, maxD, minD, gammaC, gammaP, consistent, d_disp, nullptr, s, true);
    ScratchOrder order(c, s);
    if ((rc = c.altdisp.reserve((size_t)(e1 - e0) * W * 2))) return rc;
    rc = asw_device_impl(c, dL, dR, H, W, e0, e1 - e0, win, maxD, minD, gammaC, gammaP, consistent, (int16_t *)c.altdisp.ptr, nullptr, s, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_disp, (const int16_t *)c.altdisp.ptr + (size_t)(row0 - e0) * W, (size_t)rows * W * 2, hipMemcpyDeviceToDevice, s));
    return SSAMD_OK;
}

// ------------------------------------------------------------ GSW
bool gsw_layout(GswGeom &g, int win, int XG, int DG, int Ty, size_t limit, int Hy = 1)
{
    const int p = win / 2;
    g.XG = XG; g.DG = DG; g.Ty = Ty; g.Rd = Ty == 2 ? 4 : 8; g.Hy = Hy;
    g.Tx = GSW_RX * XG; g.Dc = g.Rd * DG;
    g.threads = round_up(XG * DG, 64);
    g.nL = g.Tx + 2 * p;
    g.nT = g.nL + g.Dc - 1;
    int P = 1;
    while (8 * P < g.Dc) P <<= 1;
    g.Se = 8 * P;                                  // floats per e row (slots of 8 disparities)
    g.Ses = 3;
    while ((1 << g.Ses) < g.Se) ++g.Ses;
    g.emask = std::min(P, 32) - 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return (int)o; };
    g.off_w = take((size_t)Ty * Hy * win * g.Tx * 4);
    const int nL4 = round_up(g.nL, 4);                 // the e tasks cover 4 columns
    g.off_e = take((size_t)nL4 * g.Se * 4);
    g.off_ref = take((size_t)nL4 * 16 * 2);            // pixel staging is double-buffered (prefetch of the next image row)
    g.off_tgt = take((size_t)(g.nT + nL4 - g.nL) * 16 * 2);
    g.off_best = take((size_t)Ty * Hy * g.Tx * 8);
    g.off_cen = take((size_t)Ty * Hy * g.Tx * 4);
    g.lds_bytes = (int)off;
    return off <= limit && g.threads * Hy <= GSW_MAX_THREADS;
}

// Launch geometry of the GSW kernel: strip height Ty, XG x DG thread grid.  Relative cost model of one
// strip, per thread: every image row of the strip pays the e tile once (c_e per element), every
// (output row, window row) pair pays its weights (c_w per element) and its taps (c_tap per cell).
int gsw_search_geometry(GswGeom &best, int W, int rows, int win, int nD);

std::map<std::array<int, 4>, GswGeom> g_gsw_geom_cache;
std::map<std::array<int, 4>, bool> g_gsw_geom_tuned;      // shapes whose cached geometry was picked by measurement (gsw_device_impl)

int gsw_choose_geometry(GswGeom &best, int W, int rows, int win, int nD)      // cached like asw_choose_geometry
{
    if (!tune().gsw_geom.empty()) return gsw_search_geometry(best, W, rows, win, nD);
    std::lock_guard<std::mutex> glk(g_geom_mutex);
    const std::array<int, 4> key{W, rows, win, nD};
    auto it = g_gsw_geom_cache.find(key);
    if (it != g_gsw_geom_cache.end()) { best = it->second; return SSAMD_OK; }
    const int rc = gsw_search_geometry(best, W, rows, win, nD);
    if (rc == SSAMD_OK) {
        if (g_gsw_geom_cache.size() > 256) { g_gsw_geom_cache.clear(); g_gsw_geom_tuned.clear(); }
        g_gsw_geom_cache[key] = best;
    }
    return rc;
}

// Autotuning candidates (round 4).  The cost model above is calibrated on config 4 (193 disparities) and is up to 38 % off for
// small ranges -- the class default of StereoGSW is maxDisparity = 16 -- where narrow tiles with ONE wave per thread group and
// four-row strips win (1080p / win 11: D 0..16 2.09 -> 1.40 ms with "10,5,2,2", D 0..7 1.91 -> 1.18 ms with "16,2,2,2", D 0..32
// 2.55 -> 1.97 ms with "14,9,2,2"; profiles/r04_gsw_geometry_small_ranges.txt).  Candidates: the model's choice first, then for
// strips of 2 / 4 / 8 rows the tiles whose thread groups fill whole waves (XG x DG just below 64, 128, ... 512 lanes).
void gsw_candidates(std::vector<GswGeom> &out, const GswGeom &model, int W, int rows, int win, int nD)
{
    out.clear();
    out.push_back(model);
    const int Ty = 2, Rd = 4;
    if (rows < 2) return;
    for (int nch = model.nchunks; nch <= model.nchunks + 1 && nch <= nD; ++nch) {
        const int per = (nD + nch - 1) / nch, DG = round_up(per, Rd) / Rd;
        if (DG > 64 || (nD + DG * Rd - 1) / (DG * Rd) != nch) continue;
        for (int Hy : {1, 2, 4}) {
            if (Ty * Hy > std::max(rows, 2)) break;
            for (int T : {32, 64, 128, 192, 256, 384, 512}) {
                if (T * Hy > GSW_MAX_THREADS) break;
                for (int trim = 0; trim < 2; ++trim) {          // ... and a sixth narrower (smaller LDS slice: one more resident workgroup)
                    const int XG = std::min((T / DG) * (6 - trim) / 6, (W + GSW_RX - 1) / GSW_RX);
                    if (XG < 2) continue;
                    GswGeom g;
                    if (!gsw_layout(g, win, XG, DG, Ty, 160 * 1024, Hy)) continue;
                    g.nchunks = nch;
                    bool dup = false;
                    for (const GswGeom &o : out) dup = dup || (o.XG == g.XG && o.DG == g.DG && o.Ty == g.Ty && o.Hy == g.Hy && o.nchunks == g.nchunks);
                    if (!dup && out.size() < 36) out.push_back(g);
                }
            }
        }
    }
}

int gsw_search_geometry(GswGeom &best, int W, int rows, int win, int nD)
{
    if (!tune().gsw_geom.empty()) {                             // experiment hook: "XG,DG,Ty"
        const char *const env = tune().gsw_geom.c_str();
        int XG = 0, DG = 0, Ty = 1, Hy = 1;                       // "XG,DG[,Ty[,Hy]]"
        if (sscanf(env, "%d,%d,%d,%d", &XG, &DG, &Ty, &Hy) >= 2 && XG >= 1 && DG >= 1 && (Ty == 1 || Ty == 2) && Hy >= 1 && Hy <= 8 &&
            XG * DG <= GSW_MAX_THREADS && gsw_layout(best, win, XG, DG, Ty, 160 * 1024, Hy)) {
            best.nchunks = (nD + best.Dc - 1) / best.Dc;
            return SSAMD_OK;
        }
        return fail(SSAMD_EINVAL, "SSAMD_GSW_GEOM=%s is not a usable geometry", env);
    }
    const double c_tap = 5.3, c_w = 60.0, c_e = 70.0;
    double best_score = -1.0;
    bool found = false;
    for (int Ty = 1; Ty <= 2; ++Ty) {
        if (Ty > std::max(rows, 1)) break;
        const int Rd = Ty == 2 ? 4 : 8;
        for (int nch = 1; nch <= nD; ++nch) {
            const int per = (nD + nch - 1) / nch;
            const int DG = round_up(per, Rd) / Rd;
            if (DG > 64) continue;
            if ((nD + DG * Rd - 1) / (DG * Rd) != nch) continue;
            const int xg_cap = std::min(GSW_MAX_THREADS / DG, (W + GSW_RX - 1) / GSW_RX);
            // Hy thread groups share the e tile and the staged pixels of an image row (round 3): strips of Ty * Hy rows.
            // Built, bit-exact (SSAMD_GSW_GEOM="XG,DG,Ty,Hy", tests/test_gpu_gsw.py) and MEASURED at 1080p / D 0..192:
            // 10,25,2,2 (40-column tiles, four-row strips) 9.24 ms against 9.16 ms for 20,25,2,1 -- the third fewer e
            // elements are paid back by the narrower tile (profiles/r03_gsw_*.txt), so the search keeps Hy = 1.
            for (int Hy = 1; Hy <= 1; Hy *= 2)
            for (int XG = xg_cap; XG >= 1; --XG) {
                GswGeom g;
                if (!gsw_layout(g, win, XG, DG, Ty, 160 * 1024, Hy)) continue;
                g.nchunks = nch;
                const int tot = g.threads * Hy, TyS = Ty * Hy;
                const int waves = tot / 64, per_simd = (waves + 3) / 4;
                const int k = std::min({4 / per_simd, (160 * 1024) / g.lds_bytes, 8});   // <= 128 VGPRs: 4 waves per SIMD
                if (k < 1) continue;
                const double M = (double)win * GSW_RX * Rd * c_tap;                       // a thread aggregates its group's Ty rows only
                const double Bw = (double)((g.Tx * win + tot - 1) / tot) * c_w;           // weights and e tiles are built by all threads
                const double Be = (double)((g.nL * g.Dc + tot - 1) / tot) * c_e;
                const double strip = (double)(win + TyS - 1) * Be + (double)TyS * win * Bw + (double)Ty * win * M;
                const double eff = (double)Ty * win * M / strip;
                const double d_util = (double)nD / ((double)nch * g.Dc);
                const int xt = (W + g.Tx - 1) / g.Tx, yt = (std::max(rows, 1) + TyS - 1) / TyS;
                const double x_util = (double)W / ((double)xt * g.Tx);
                const double y_util = (double)std::max(rows, 1) / ((double)yt * TyS);
                const double nwg = (double)xt * yt * nch, slots = 256.0 * k;
                const double tail = nwg / (std::ceil(nwg / slots) * slots);
                const double score = (double)k * XG * DG * Hy * eff * d_util * x_util * y_util * tail;
                if (score > best_score) { best_score = score; best = g; found = true; }
            }
            if (DG <= 1) break;
        }
    }
    return found ? SSAMD_OK : fail(SSAMD_ELIMIT, "no GSW launch geometry fits LDS for winSize=%d nD=%d", win, nD);
}

// support weight as a function of the integer squared colour distance, in the reference's
// arithmetic: fl32 distance (sqrt in double), float division by gamma, float exp (_passive.cpp:457-463, 495)
int get_gsw_table(Ctx &c, int gamma, hipStream_t s, const float **out)
{
    if (TableEntry *e = c.gswTabs.find(gamma, 0.0)) { *out = (const float *)e->dev.ptr; return SSAMD_OK; }
    if (c.gswTabs.entries.size() >= c.gswTabs.max_entries) {
        HIP_TRY(hipDeviceSynchronize());         // see get_prox
        (void)hipFree(c.gswTabs.entries.back().dev.ptr);
        c.gswTabs.entries.pop_back();
    }
    c.gswTabs.entries.emplace_front();
    TableEntry &e = c.gswTabs.entries.front();
    e.k0 = gamma; e.k1 = 0.0;
    e.host.resize(GSW_TAB_SIZE);
    for (int v = 0; v < GSW_TAB_SIZE; ++v) {
        const float dist = (float)(0.0f + std::sqrt((double)v));
        e.host[v] = expf(-dist / gamma);
    }
    int rc = e.dev.reserve(e.host.size() * 4);
    if (rc) { c.gswTabs.entries.pop_front(); return rc; }
    hipError_t he = hipMemcpyAsync(e.dev.ptr, e.host.data(), e.host.size() * 4, hipMemcpyHostToDevice, s);
    if (he != hipSuccess) {
        (void)hipFree(e.dev.ptr);
        c.gswTabs.entries.pop_front();
        return fail(SSAMD_EHIP, "hipMemcpyAsync(GSW weight table) failed: %s", hipGetErrorString(he));
    }
    *out = (const float *)e.dev.ptr;
    return SSAMD_OK;
}

int gsw_device_impl(Ctx &c, const uint8_t *dL, const uint8_t *dR, int H, int W, int row0, int rows, int win, int maxD,
                    int minD, int gamma, float fMax, int iterations, int16_t *d_disp, hipStream_t s, const RemapSrc *rm = nullptr)
{
    int rc = check_common(H, W, win, minD, maxD, row0, rows);
    if (rc) return rc;
    if (gamma == 0) return fail(SSAMD_EINVAL, "gamma must be non-zero");
    if (rows == 0) return SSAMD_OK;
    ScratchOrder order(c, s);
    const int p = win / 2, nD = maxD - minD + 1;
    const size_t npix = (size_t)H * W, nout = (size_t)rows * W;
    if ((rc = c.keyL.reserve(nout * 8)) || (rc = c.keyR.reserve(nout * 8))) return rc;
    HIP_TRY(hipMemsetAsync(c.keyL.ptr, 0xFF, nout * 8, s));
    HIP_TRY(hipMemsetAsync(c.keyR.ptr, 0xFF, nout * 8, s));
    if (nD >= 1) {
        // packed pixels live in the (larger) ASW record buffers: 4 B/pixel
        if ((rc = c.recL.reserve(npix * 4)) || (rc = c.recR.reserve(npix * 4))) return rc;
        const float *d_tab = nullptr;
        if ((rc = get_gsw_table(c, gamma, s, &d_tab))) return rc;
        const int r0 = std::max(0, row0 - p), r1 = std::min(H, row0 + rows + p);
        const long long np = (long long)(r1 - r0) * W;
        const int blocks = (int)std::min<long long>((np + 255) / 256, 256 * 8);
        {
            Timed t(c, s, SSAMD_K_LAB);
            if (rm) {       // raw frames through the rig's maps: rectification + packing of both images, one launch
                hipLaunchKernelGGL(remap_pack_pair_kernel, dim3(blocks), dim3(256), 0, s, *rm, (uint32_t *)c.recL.ptr, (uint32_t *)c.recR.ptr,
                                   (long long)r0 * W, np);
            } else {
                hipLaunchKernelGGL(bgr_pack_kernel, dim3(blocks), dim3(256), 0, s, dL + (size_t)r0 * W * 3,
                                   (uint32_t *)c.recL.ptr + (size_t)r0 * W, np);
                hipLaunchKernelGGL(bgr_pack_kernel, dim3(blocks), dim3(256), 0, s, dR + (size_t)r0 * W * 3,
                                   (uint32_t *)c.recR.ptr + (size_t)r0 * W, np);
            }
            HIP_TRY(hipGetLastError());
        }
        GswArgs a;
        if ((rc = gsw_choose_geometry(a.g, W, rows, win, nD))) return rc;
        a.tab = d_tab;
        a.H = H; a.W = W; a.win = win; a.pad = p; a.minD = minD; a.maxD = maxD; a.row0 = row0; a.rows = rows;
        a.iterations = iterations; a.fMax = fMax;
        // both passes with geometry g (the winner-take-all keys merge by atomicMin: repeated launches are idempotent)
        auto launch = [&](const GswGeom &g) -> int {
            a.g = g;
            const int TyS = g.Ty * g.Hy;
            const dim3 grid((W + g.Tx - 1) / g.Tx, (rows + TyS - 1) / TyS, g.nchunks), block(g.threads * g.Hy);
            auto kernel = g.Ty == 2 ? gsw_aggregate_kernel<2, 4> : gsw_aggregate_kernel<1, 8>;
            if (int grc = grant_dyn_lds(c, (const void *)kernel, g.lds_bytes)) return grc;
            for (int pass = 0; pass < 2; ++pass) {
                a.right = pass;
                a.ref = (const uint32_t *)(pass ? c.recR.ptr : c.recL.ptr);
                a.tgt = (const uint32_t *)(pass ? c.recL.ptr : c.recR.ptr);
                a.key = (u64 *)(pass ? c.keyR.ptr : c.keyL.ptr);
                hipLaunchKernelGGL(kernel, grid, block, g.lds_bytes, s, a);
                HIP_TRY(hipGetLastError());
            }
            return SSAMD_OK;
        };
        // Autotuning (ssamd_autotune, as for ASW): the first call of a shape times the candidates of gsw_candidates on the
        // call's own buffers -- three rounds in round-robin order, fastest launch of each -- and caches the winner.  Default
        // mode: calls of at most 6e10 window taps (both passes; 2 lane-ops each: about 5 ms), where the model is least reliable.
        {
            const std::array<int, 4> shape{W, rows, win, nD};
            const double call_taps = 2.0 * (double)W * rows * nD * win * win;
            const int tune_mode = g_autotune.load();
            bool tuned_already;
            { std::lock_guard<std::mutex> glk(g_geom_mutex); tuned_already = g_gsw_geom_tuned.count(shape) != 0; }
            if ((tune_mode > 0 || (tune_mode < 0 && call_taps <= ASW_AUTOTUNE_SMALL_TAPS)) && tune().gsw_geom.empty() && !tuned_already) {
                std::vector<GswGeom> trial;
                gsw_candidates(trial, a.g, W, rows, win, nD);
                if (trial.size() >= 2) {
                    hipEvent_t e0 = nullptr, e1 = nullptr;
                    HIP_TRY(hipEventCreate(&e0));
                    if (hipEventCreate(&e1) != hipSuccess) {
                        (void)hipEventDestroy(e0);
                        return fail(SSAMD_EHIP, "hipEventCreate failed");
                    }
                    std::vector<float> cand_ms(trial.size(), 3.0e38f), warm_ms(trial.size(), 3.0e38f);
                    std::vector<char> alive(trial.size(), 1);
                    for (int round = 0; round < 3; ++round) {
                        for (size_t ci = 0; ci < trial.size(); ++ci) {
                            if (!alive[ci]) continue;
                            float ms = 3.0e38f;
                            if (hipEventRecord(e0, s) == hipSuccess && launch(trial[ci]) == SSAMD_OK && hipEventRecord(e1, s) == hipSuccess &&
                                hipEventSynchronize(e1) == hipSuccess)
                                (void)hipEventElapsedTime(&ms, e0, e1);
                            if (round > 0) cand_ms[ci] = std::min(cand_ms[ci], ms);      // round 0 is warm-up
                            else warm_ms[ci] = ms;
                        }
                        if (round == 0) {
                            // a candidate that is twice as slow as the best one in the warm-up round (very narrow tiles can be several
                            // times slower than the model's choice) is not timed again: bounds what the first call of a shape costs
                            const float wbest = *std::min_element(warm_ms.begin(), warm_ms.end());
                            for (size_t ci = 1; ci < trial.size(); ++ci) alive[ci] = warm_ms[ci] <= 2.0f * wbest;
                        }
                    }
                    (void)hipGetLastError();
                    g_err.clear();            // a failed trial launch is not the call's error
                    GswGeom fastest = trial[0];
                    float best_ms = 3.0e38f;
                    for (size_t ci = 0; ci < trial.size(); ++ci)      // the model's own choice (first) keeps the job unless another is clearly faster
                        if (cand_ms[ci] < best_ms * (ci == 0 ? 1.0f : 0.985f)) { best_ms = cand_ms[ci]; fastest = trial[ci]; }
                    (void)hipEventDestroy(e0);
                    (void)hipEventDestroy(e1);
                    std::lock_guard<std::mutex> glk(g_geom_mutex);
                    g_gsw_geom_cache[shape] = fastest;
                    g_gsw_geom_tuned[shape] = true;
                    a.g = fastest;
                }
            }
        }
        {
            const GswGeom final_geom = a.g;
            Timed t(c, s, SSAMD_K_GSW_AGG);
            if ((rc = launch(final_geom))) return rc;
        }
    }
    return launch_finalize(c, SSAMD_K_GSW_FIN, true, rows, W, d_disp, s);   // consistency is unconditional in GSW
}

}  // namespace

// =================================================================== C ABI
namespace {

// ---- host-buffer paths: H2D copy of rows [in0, in1) of both images, kernels on output rows [o0, o1), D2H copy.
// Used for the whole image (ssamd_asw / ssamd_gsw) and, one host thread per device, for the row strips of
// ssamd_*_multi: output row y needs input rows y-pad .. y+pad only and the left-right check and occlusion filling
// are row-local (_passive.cpp:38-40, 60-62, 251-285), so a strip that carries its halo reproduces the rows of the
// whole-image result bit for bit.
struct HostJob {
    const uint8_t *img1, *img2;
    int H, W, win, maxD, minD;
    int o0, o1;                    // output rows of the full image
    int16_t *disparity;            // full-image output [H][W]
    // ASW
    double gammaC, gammaP; int consistent; float *costs; bool alternate;
    bool exact;                    // fp64 tie-break pass (ssamd_asw_exact*)
    int16_t *raw_right;            // verification dump (ssamd_asw_argmins): raw right-referenced matches, full image
    // GSW
    int gamma; float fMax; int iterations;
};

int asw_host_rows(const HostJob &j, int device)
{
    CtxLock c;
    int rc = get_ctx(device, c);
    if (rc) return rc;
    if ((rc = check_common(j.H, j.W, j.win, j.minD, j.maxD, j.o0, j.o1 - j.o0))) return rc;
    const int p = j.win / 2 + (j.alternate ? 1 : 0);        // alternate rows: + the exact row beyond an odd first / last row
    const int in0 = std::max(0, j.o0 - p), in1 = std::min(j.H, j.o1 + p), rows = j.o1 - j.o0;
    const size_t nb = (size_t)(in1 - in0) * j.W * 3, nout = (size_t)rows * j.W;
    if ((rc = c->imgL.reserve(nb)) || (rc = c->imgR.reserve(nb)) || (rc = c->disp.reserve(nout * 2))) return rc;
    const size_t ncost = j.costs ? nout * (size_t)std::max(1, j.maxD - j.minD + 1) : 0;
    if (j.costs && (rc = c->costs.reserve(ncost * 4))) return rc;
    if (j.raw_right && (rc = c->lab.reserve(nout * 2))) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(c->imgL.ptr, j.img1 + (size_t)in0 * j.W * 3, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->imgR.ptr, j.img2 + (size_t)in0 * j.W * 3, nb, hipMemcpyHostToDevice, s));
    if (j.costs) HIP_TRY(hipMemsetAsync(c->costs.ptr, 0xFF, ncost * 4, s));      // 0xFFFFFFFF = NaN
    if (j.alternate)
        rc = asw_alternate_rows(*c, (const uint8_t *)c->imgL.ptr, (const uint8_t *)c->imgR.ptr, in1 - in0, j.W, j.o0 - in0, rows, in0 & 1,
                                j.win, j.maxD, j.minD, j.gammaC, j.gammaP, j.consistent, (int16_t *)c->disp.ptr, s);
    else
        rc = asw_device_impl(*c, (const uint8_t *)c->imgL.ptr, (const uint8_t *)c->imgR.ptr, in1 - in0, j.W, j.o0 - in0, rows,
                             j.win, j.maxD, j.minD, j.gammaC, j.gammaP, j.consistent, (int16_t *)c->disp.ptr,
                             j.costs ? (float *)c->costs.ptr : nullptr, s, false,
                             j.raw_right ? (int16_t *)c->lab.ptr : nullptr, nullptr, j.exact);
    if (rc) return rc;
    if (j.raw_right)
        HIP_TRY(hipMemcpyAsync(j.raw_right + (size_t)j.o0 * j.W, c->lab.ptr, nout * 2, hipMemcpyDeviceToHost, s));
    if (j.disparity)
        HIP_TRY(hipMemcpyAsync(j.disparity + (size_t)j.o0 * j.W, c->disp.ptr, nout * 2, hipMemcpyDeviceToHost, s));
    if (j.costs) HIP_TRY(hipMemcpyAsync(j.costs, c->costs.ptr, ncost * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SSAMD_OK;
}

int gsw_host_rows(const HostJob &j, int device)
{
    CtxLock c;
    int rc = get_ctx(device, c);
    if (rc) return rc;
    if ((rc = check_common(j.H, j.W, j.win, j.minD, j.maxD, j.o0, j.o1 - j.o0))) return rc;
    const int p = j.win / 2, in0 = std::max(0, j.o0 - p), in1 = std::min(j.H, j.o1 + p), rows = j.o1 - j.o0;
    const size_t nb = (size_t)(in1 - in0) * j.W * 3, nout = (size_t)rows * j.W;
    if ((rc = c->imgL.reserve(nb)) || (rc = c->imgR.reserve(nb)) || (rc = c->disp.reserve(nout * 2))) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(c->imgL.ptr, j.img1 + (size_t)in0 * j.W * 3, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->imgR.ptr, j.img2 + (size_t)in0 * j.W * 3, nb, hipMemcpyHostToDevice, s));
    rc = gsw_device_impl(*c, (const uint8_t *)c->imgL.ptr, (const uint8_t *)c->imgR.ptr, in1 - in0, j.W, j.o0 - in0, rows,
                         j.win, j.maxD, j.minD, j.gamma, j.fMax, j.iterations, (int16_t *)c->disp.ptr, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(j.disparity + (size_t)j.o0 * j.W, c->disp.ptr, nout * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SSAMD_OK;
}

// Contiguous row strips whose heights differ by at most one row (empty when there are more devices than rows) --
// the same cut as simplestereo_amd/strips.py::strip_bounds.  One host thread per strip: each takes its own device's lock, so the
// copies and kernels of all devices overlap.  The first failing strip's code and message are returned.
int run_strips(const HostJob &job, const int *devices, int n_devices, int (*fn)(const HostJob &, int))
{
    if (!devices || n_devices < 1) return fail(SSAMD_EINVAL, "devices must name at least one GPU");
    if (n_devices > 16) return fail(SSAMD_EINVAL, "at most 16 devices");
    for (int a = 0; a < n_devices; ++a) {
        if (devices[a] < 0) return fail(SSAMD_EINVAL, "devices[%d] = %d: explicit non-negative ordinals only", a, devices[a]);
        // test hook SSAMD_MULTI_ALLOW_REPEAT: a 1-GPU box exercises the strip cut with one device listed several
        // times (the strips then simply queue on that device's mutex)
        for (int b = 0; b < a && !tune().multi_allow_repeat; ++b)
            if (devices[a] == devices[b]) return fail(SSAMD_EINVAL, "device %d listed twice", devices[a]);
    }
    int rc = check_common(job.H, job.W, job.win, job.minD, job.maxD, 0, job.H);
    if (rc) return rc;
    const int base = job.H / n_devices, extra = job.H % n_devices;
    std::vector<int> codes(n_devices, SSAMD_OK);
    std::vector<std::string> msgs(n_devices);
    std::vector<std::thread> th;
    for (int k = 0; k < n_devices; ++k) {
        HostJob j = job;
        j.o0 = k * base + std::min(k, extra);
        j.o1 = j.o0 + base + (k < extra ? 1 : 0);
        if (j.o1 <= j.o0) continue;                      // more devices than rows: nothing for this one
        const int dev = devices[k];
        try {
            th.emplace_back([j, dev, k, fn, &codes, &msgs]() {
                codes[k] = fn(j, dev);
                if (codes[k]) msgs[k] = g_err;           // thread-local message of the worker
            });
        } catch (const std::exception &e) {              // no exception may cross the C ABI: finish what runs, report
            codes[k] = SSAMD_ENOMEM;
            msgs[k] = std::string("could not start a host thread: ") + e.what();
            break;
        }
    }
    for (auto &t : th) t.join();
    for (int k = 0; k < n_devices; ++k)
        if (codes[k]) return fail(codes[k], "strip %d on device %d: %s", k, devices[k], msgs[k].c_str());
    return SSAMD_OK;
}

HostJob asw_job(const uint8_t *img1, const uint8_t *img2, int H, int W, int win, int maxD, int minD, double gammaC,
                double gammaP, int consistent, int16_t *disparity, float *costs, bool alternate)
{
    HostJob j{};
    j.img1 = img1; j.img2 = img2; j.H = H; j.W = W; j.win = win; j.maxD = maxD; j.minD = minD; j.o0 = 0; j.o1 = H;
    j.disparity = disparity; j.gammaC = gammaC; j.gammaP = gammaP; j.consistent = consistent; j.costs = costs;
    j.alternate = alternate;
    return j;
}

HostJob gsw_job(const uint8_t *img1, const uint8_t *img2, int H, int W, int win, int maxD, int minD, int gamma, float fMax,
                int iterations, int16_t *disparity)
{
    HostJob j{};
    j.img1 = img1; j.img2 = img2; j.H = H; j.W = W; j.win = win; j.maxD = maxD; j.minD = minD; j.o0 = 0; j.o1 = H;
    j.disparity = disparity; j.gamma = gamma; j.fMax = fMax; j.iterations = iterations;
    return j;
}

}  // namespace

extern "C" {

int ssamd_abi_version(void) { return SSAMD_ABI_VERSION; }
const char *ssamd_last_error(void) { return g_err.c_str(); }

int ssamd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ssamd_kernel_name(int slot)
{
    static const char *names[SSAMD_K_COUNT] = {"asw pre-pass (asw_prepass_kernel: Lab records + TAD volume; or bgr2lab_records_pair_kernel, asw_tad_volume_kernel)", "asw aggregation kernel (asw_aggregate_pipe / _wave / asw_aggregate_kernel)",
                                               "asw finalize (wta_decode / lr_check_fill)",
                                               "gsw_aggregate_kernel", "gsw finalize (lr_check_fill)", "remap_bgr_kernel", "reproject_kernel",
                                               "asw_alt_fill_kernel", "asw fp64 tie-break pass (bgr2lab_f64_pair + asw_exact_winners / _eval / _resolve / _patch kernels)"};
    return (slot >= 0 && slot < SSAMD_K_COUNT) ? names[slot] : "";
}

int ssamd_set_option(const char *name, const char *value)
{
    if (!name) return fail(SSAMD_EINVAL, "option name is NULL");
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    if (value) {
        if (!tuning_assign(g_tuning, name, value)) return fail(SSAMD_EINVAL, "unknown option %s", name);
    } else {
        // NULL = back to the value the library loaded from the environment (not "unset": a process started with
        // SSAMD_ASW_WAVE=0 keeps that setting after a `with options(...)` block of a test)
        const auto it = g_tuning_env.find(name);
        if (!tuning_assign(g_tuning, name, it == g_tuning_env.end() ? nullptr : it->second.c_str()))
            return fail(SSAMD_EINVAL, "unknown option %s", name);
    }
    if (std::string(name) == "SSAMD_AUTOTUNE")
        g_autotune.store(g_tuning.autotune_env != -2 ? g_tuning.autotune_env : -1);
    g_tune_version.fetch_add(1, std::memory_order_release);
    return SSAMD_OK;
}

int ssamd_counter(int device, const char *name, long long *value)
{
    if (!name || !value) return fail(SSAMD_EINVAL, "name / value is NULL");
    CtxLock c;
    int rc = get_ctx(device, c);
    if (rc) return rc;
    const std::string n(name);
    if (n == "evol_fallbacks") *value = c->evol_fallbacks;
    else if (n == "evol_bytes") *value = (long long)c->evol.cap;
    else if (n == "tail_splits") *value = c->tail_splits;
    else if (n == "exact_calls") *value = c->exact_calls;
    else if (n == "static_tile_mismatch") *value = c->static_tile_mismatch;
    else if (n == "exact_entries" || n == "exact_flagged_left" || n == "exact_flagged_right" || n == "exact_overflow" || n == "exact_raw_entries") {
        // of the LAST exact call on this device: candidates re-evaluated in fp64, pixels with near-ties, whether the queue overflowed
        unsigned int ctr[5] = {0, 0, 0, 0, 0};
        if (c->xflags.ptr && c->exact_calls > 0) {       // (the counters are the first 64 bytes of the flag buffer: one memset per call)
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(ctr, c->xflags.ptr, sizeof(ctr), hipMemcpyDeviceToHost));
        }
        *value = n == "exact_entries" ? ctr[0] : n == "exact_flagged_left" ? ctr[1] : n == "exact_flagged_right" ? ctr[2] :
                 n == "exact_raw_entries" ? ctr[4] : ((ctr[0] > c->xcap || (c->xrawcap && ctr[4] > c->xrawcap)) ? 1 : 0);
    }
    else return fail(SSAMD_EINVAL, "unknown counter %s", name);
    return SSAMD_OK;
}

int ssamd_autotune(int on)
{
    return g_autotune.exchange(on > 0 ? 1 : (on < 0 ? -1 : 0));
}

int ssamd_asw_geometry(int width, int rows, int winSize, int maxDisparity, int minDisparity, int *out)
{
    if (!out) return fail(SSAMD_EINVAL, "out is NULL");
    int rc = check_common(1 << 14, width, winSize, minDisparity, maxDisparity, 0, 0);
    if (rc) return rc;
    const int nD = maxDisparity - minDisparity + 1;
    if (nD < 1) return fail(SSAMD_EINVAL, "empty disparity range");
    AswGeom g;
    if ((rc = asw_choose_geometry(g, width, rows, winSize, nD))) return rc;
    out[0] = g.Tx; out[1] = g.Dc; out[2] = g.nchunks; out[3] = g.threads; out[4] = g.lds_bytes;
    out[5] = (width + g.Tx - 1) / g.Tx; out[6] = rows; out[7] = g.nchunks;
    AswWaveGeom wg;
    if (g.wave_rx && asw_wave_layout(wg, winSize, nD, g.wave_rx, tune().wave_unroll != 0)) {      // a "tile" = the four strips of a workgroup's waves (LDS of the plain call: no cost dump)
        out[0] = wg.Txw * wg.waves; out[1] = wg.Dc; out[2] = 1; out[3] = 64 * wg.waves; out[4] = wg.wave_lds * wg.waves;
        out[5] = (width + out[0] - 1) / out[0]; out[7] = 1;
    }
    return SSAMD_OK;
}

int ssamd_asw_kernel_form(int width, int rows, int winSize, int maxDisparity, int minDisparity, int *out)
{
    if (!out) return fail(SSAMD_EINVAL, "out is NULL");
    int rc = check_common(1 << 14, width, winSize, minDisparity, maxDisparity, 0, 0);
    if (rc) return rc;
    const int nD = maxDisparity - minDisparity + 1;
    if (nD < 1) return fail(SSAMD_EINVAL, "empty disparity range");
    AswGeom g;
    if ((rc = asw_choose_geometry(g, width, rows, winSize, nD))) return rc;
    out[0] = g.pipe; out[1] = g.Rx; out[2] = g.JC >= winSize ? 0 : g.JC; out[3] = g.pipe ? g.dephase : 0;
    out[4] = g.wave_rx & 15;
    if (g.wave_rx) { out[0] = 0; out[1] = g.wave_rx & 15; out[2] = 0; out[3] = 0; }
    return SSAMD_OK;
}

int ssamd_gsw_geometry(int width, int rows, int winSize, int maxDisparity, int minDisparity, int *out)
{
    if (!out) return fail(SSAMD_EINVAL, "out is NULL");
    int rc = check_common(1 << 14, width, winSize, minDisparity, maxDisparity, 0, 0);
    if (rc) return rc;
    const int nD = maxDisparity - minDisparity + 1;
    if (nD < 1) return fail(SSAMD_EINVAL, "empty disparity range");
    GswGeom g;
    if ((rc = gsw_choose_geometry(g, width, rows, winSize, nD))) return rc;
    out[0] = g.Tx; out[1] = g.Dc; out[2] = g.nchunks; out[3] = g.threads * g.Hy; out[4] = g.lds_bytes;
    out[5] = (width + g.Tx - 1) / g.Tx; out[6] = (rows + g.Ty * g.Hy - 1) / (g.Ty * g.Hy); out[7] = g.nchunks; out[8] = g.Ty * g.Hy;
    return SSAMD_OK;
}

int ssamd_asw_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                     int winSize, int maxDisparity, int minDisparity, double gammaC, double gammaP, int consistent,
                     int16_t *d_disparity, void *stream)
{
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    return asw_device_impl(*c, d_img1, d_img2, height, width, out_row0, out_rows, winSize, maxDisparity, minDisparity,
                           gammaC, gammaP, consistent, d_disparity, nullptr, (hipStream_t)stream);
}

int ssamd_asw_device_rows2(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                           int skip_row0, int skip_rows, int winSize, int maxDisparity, int minDisparity, double gammaC, double gammaP,
                           int consistent, int16_t *d_disparity, void *stream)
{
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    if (skip_rows < 0 || (skip_rows > 0 && (skip_row0 < out_row0 || skip_row0 + skip_rows > out_row0 + out_rows)))
        return fail(SSAMD_EINVAL, "the skipped rows [%d,%d) must lie inside the output rows [%d,%d)", skip_row0, skip_row0 + skip_rows,
                    out_row0, out_row0 + out_rows);
    return asw_device_impl(*c, d_img1, d_img2, height, width, out_row0, out_rows, winSize, maxDisparity, minDisparity, gammaC, gammaP,
                           consistent, d_disparity, nullptr, (hipStream_t)stream, false, nullptr, nullptr, false,
                           skip_rows > 0 ? skip_row0 - out_row0 : 0, skip_rows);
}

int ssamd_asw_exact_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                           int winSize, int maxDisparity, int minDisparity, double gammaC, double gammaP, int consistent,
                           int16_t *d_disparity, void *stream)
{
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    return asw_device_impl(*c, d_img1, d_img2, height, width, out_row0, out_rows, winSize, maxDisparity, minDisparity,
                           gammaC, gammaP, consistent, d_disparity, nullptr, (hipStream_t)stream, false, nullptr, nullptr, true);
}

int ssamd_asw_exact_device_rows2(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                                 int skip_row0, int skip_rows, int winSize, int maxDisparity, int minDisparity, double gammaC, double gammaP,
                                 int consistent, int16_t *d_disparity, void *stream)
{
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    if (skip_rows < 0 || (skip_rows > 0 && (skip_row0 < out_row0 || skip_row0 + skip_rows > out_row0 + out_rows)))
        return fail(SSAMD_EINVAL, "the skipped rows [%d,%d) must lie inside the output rows [%d,%d)", skip_row0, skip_row0 + skip_rows,
                    out_row0, out_row0 + out_rows);
    return asw_device_impl(*c, d_img1, d_img2, height, width, out_row0, out_rows, winSize, maxDisparity, minDisparity, gammaC, gammaP,
                           consistent, d_disparity, nullptr, (hipStream_t)stream, false, nullptr, nullptr, true,
                           skip_rows > 0 ? skip_row0 - out_row0 : 0, skip_rows);
}

int ssamd_asw_exact_rectified_device(const uint8_t *d_raw1, const uint8_t *d_raw2, int src_height, int src_width,
                                     const float *d_mapx1, const float *d_mapy1, const float *d_mapx2, const float *d_mapy2,
                                     int height, int width, int interpolation, int winSize, int maxDisparity, int minDisparity,
                                     double gammaC, double gammaP, int consistent, int16_t *d_disparity, void *stream)
{
    if (!d_raw1 || !d_raw2 || !d_mapx1 || !d_mapy1 || !d_mapx2 || !d_mapy2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    if (src_height <= 0 || src_width <= 0) return fail(SSAMD_EINVAL, "Wrong image dimensions!");
    if (interpolation != 0 && interpolation != 1) return fail(SSAMD_EINVAL, "only INTER_NEAREST (0) and INTER_LINEAR (1) are supported");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    RemapSrc rm;
    rm.src1 = d_raw1; rm.src2 = d_raw2; rm.mapx1 = d_mapx1; rm.mapy1 = d_mapy1; rm.mapx2 = d_mapx2; rm.mapy2 = d_mapy2;
    rm.Hs = src_height; rm.Ws = src_width; rm.nearest = interpolation == 0 ? 1 : 0;
    return asw_device_impl(*c, nullptr, nullptr, height, width, 0, height, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent,
                           d_disparity, nullptr, (hipStream_t)stream, false, nullptr, &rm, true);
}

int ssamd_asw_exact(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                    int minDisparity, double gammaC, double gammaP, int consistent, int16_t *disparity, int device)
{
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    HostJob j = asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent, disparity, nullptr, false);
    j.exact = true;
    return asw_host_rows(j, device);
}

int ssamd_asw_rectified_device(const uint8_t *d_raw1, const uint8_t *d_raw2, int src_height, int src_width,
                               const float *d_mapx1, const float *d_mapy1, const float *d_mapx2, const float *d_mapy2,
                               int height, int width, int interpolation, int winSize, int maxDisparity, int minDisparity,
                               double gammaC, double gammaP, int consistent, int16_t *d_disparity, void *stream)
{
    if (!d_raw1 || !d_raw2 || !d_mapx1 || !d_mapy1 || !d_mapx2 || !d_mapy2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    if (src_height <= 0 || src_width <= 0) return fail(SSAMD_EINVAL, "Wrong image dimensions!");
    if (interpolation != 0 && interpolation != 1) return fail(SSAMD_EINVAL, "only INTER_NEAREST (0) and INTER_LINEAR (1) are supported");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    RemapSrc rm;
    rm.src1 = d_raw1; rm.src2 = d_raw2; rm.mapx1 = d_mapx1; rm.mapy1 = d_mapy1; rm.mapx2 = d_mapx2; rm.mapy2 = d_mapy2;
    rm.Hs = src_height; rm.Ws = src_width; rm.nearest = interpolation == 0 ? 1 : 0;
    return asw_device_impl(*c, nullptr, nullptr, height, width, 0, height, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent,
                           d_disparity, nullptr, (hipStream_t)stream, false, nullptr, &rm);
}

int ssamd_asw(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
              int minDisparity, double gammaC, double gammaP, int consistent, int16_t *disparity, int device)
{
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    return asw_host_rows(asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent,
                                 disparity, nullptr, false), device);
}

int ssamd_asw_multi(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                    int minDisparity, double gammaC, double gammaP, int consistent, int16_t *disparity,
                    const int *devices, int n_devices)
{
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    return run_strips(asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent,
                              disparity, nullptr, false), devices, n_devices, asw_host_rows);
}

int ssamd_asw_exact_multi(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                          int minDisparity, double gammaC, double gammaP, int consistent, int16_t *disparity,
                          const int *devices, int n_devices)
{
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    HostJob j = asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent, disparity, nullptr, false);
    j.exact = true;          // (rows are independent jobs and the tie-break pass is row-local: each strip runs its own)
    return run_strips(j, devices, n_devices, asw_host_rows);
}

int ssamd_asw_alternate(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                        int minDisparity, double gammaC, double gammaP, int consistent, int16_t *disparity, int device)
{
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    return asw_host_rows(asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent,
                                 disparity, nullptr, true), device);
}

int ssamd_asw_alternate_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int winSize,
                               int maxDisparity, int minDisparity, double gammaC, double gammaP, int consistent,
                               int16_t *d_disparity, void *stream)
{
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    return asw_device_impl(*c, d_img1, d_img2, height, width, 0, height, winSize, maxDisparity, minDisparity, gammaC,
                           gammaP, consistent, d_disparity, nullptr, (hipStream_t)stream, true);
}

int ssamd_asw_alternate_rows_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                                    int row_parity, int winSize, int maxDisparity, int minDisparity, double gammaC, double gammaP,
                                    int consistent, int16_t *d_disparity, void *stream)
{
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    if (row_parity != 0 && row_parity != 1) return fail(SSAMD_EINVAL, "row_parity must be 0 or 1");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    return asw_alternate_rows(*c, d_img1, d_img2, height, width, out_row0, out_rows, row_parity, winSize, maxDisparity, minDisparity,
                              gammaC, gammaP, consistent, d_disparity, (hipStream_t)stream);
}

int ssamd_asw_alternate_multi(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                              int minDisparity, double gammaC, double gammaP, int consistent, int16_t *disparity,
                              const int *devices, int n_devices)
{
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    return run_strips(asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, consistent,
                              disparity, nullptr, true), devices, n_devices, asw_host_rows);
}

int ssamd_asw_costs(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                    int minDisparity, double gammaC, double gammaP, float *costs, int device)
{
    if (!img1 || !img2 || !costs) return fail(SSAMD_EINVAL, "NULL buffer");
    if (maxDisparity < minDisparity) return fail(SSAMD_EINVAL, "empty disparity range");
    return asw_host_rows(asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, 0, nullptr,
                                 costs, false), device);
}

int ssamd_asw_argmins(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                      int minDisparity, double gammaC, double gammaP, int16_t *left_disparity, int16_t *right_match,
                      int device)
{
    if (!img1 || !img2 || !left_disparity || !right_match) return fail(SSAMD_EINVAL, "NULL buffer");
    if (maxDisparity < minDisparity) return fail(SSAMD_EINVAL, "empty disparity range");
    HostJob j = asw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gammaC, gammaP, 1, left_disparity,
                        nullptr, false);
    j.raw_right = right_match;
    return asw_host_rows(j, device);
}

int ssamd_bgr2lab(const uint8_t *img, int height, int width, float *lab, int device)
{
    if (!img || !lab || height <= 0 || width <= 0) return fail(SSAMD_EINVAL, "bad argument");
    CtxLock c;
    int rc = get_ctx(device, c);
    if (rc) return rc;
    const size_t npix = (size_t)height * width;
    if ((rc = c->imgL.reserve(npix * 3)) || (rc = c->lab.reserve(npix * 12))) return rc;
    hipStream_t s = c->stream;
    ScratchOrder order(*c, s);
    HIP_TRY(hipMemcpyAsync(c->imgL.ptr, img, npix * 3, hipMemcpyHostToDevice, s));
    const int blocks = (int)std::min<size_t>((npix + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(bgr2lab_f32_kernel, dim3(blocks), dim3(256), 0, s, (const uint8_t *)c->imgL.ptr,
                       (float *)c->lab.ptr, (long long)npix);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(lab, c->lab.ptr, npix * 12, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SSAMD_OK;
}

// Verification: the fp64 costs the tie-break pass computes for n given candidates (y, x, d triples), and optionally the fp64 Lab
// images it reads -- to compare with the oracle's fp64 costs bit for bit.
int ssamd_debug_exact_costs(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, double gammaC, double gammaP,
                            int n, const int *yxd, double *costs, double *lab1, double *lab2)
{
    if (!img1 || !img2 || !yxd || !costs || n < 0) return fail(SSAMD_EINVAL, "NULL buffer");
    int rc = check_common(height, width, winSize, 0, 0, 0, height);
    if (rc) return rc;
    CtxLock c;
    if ((rc = get_ctx(-1, c))) return rc;
    const size_t npix = (size_t)height * width;
    hipStream_t s = c->stream;
    ScratchOrder order(*c, s);
    if ((rc = c->imgL.reserve(npix * 3)) || (rc = c->imgR.reserve(npix * 3)) || (rc = c->recL.reserve(npix * sizeof(PixRec))) ||
        (rc = c->recR.reserve(npix * sizeof(PixRec))) || (rc = c->xlabL.reserve(npix * 24)) || (rc = c->xlabR.reserve(npix * 24)) ||
        (rc = c->xqueue.reserve((size_t)std::max(n, 1) * 8)) || (rc = c->xcost.reserve((size_t)std::max(n, 1) * 8)) || (rc = c->xctr.reserve(64)))
        return rc;
    HIP_TRY(hipMemcpyAsync(c->imgL.ptr, img1, npix * 3, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->imgR.ptr, img2, npix * 3, hipMemcpyHostToDevice, s));
    const int blocks = (int)std::min<long long>((2 * (long long)npix + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(bgr2lab_records_pair_kernel, dim3(blocks), dim3(256), 0, s, (const uint8_t *)c->imgL.ptr, (const uint8_t *)c->imgR.ptr,
                       (PixRec *)c->recL.ptr, (PixRec *)c->recR.ptr, (long long)npix);
    hipLaunchKernelGGL(bgr2lab_f64_pair_kernel, dim3(blocks), dim3(256), 0, s, (const PixRec *)c->recL.ptr, (const PixRec *)c->recR.ptr,
                       (double *)c->xlabL.ptr, (double *)c->xlabR.ptr, (long long)npix);
    std::vector<u64> ent((size_t)n);
    for (int k = 0; k < n; ++k) {
        const int y = yxd[3 * k], x = yxd[3 * k + 1], d = yxd[3 * k + 2];
        if (y < 0 || y >= height || x < 0 || x >= width || d < 0 || x - d < 0) return fail(SSAMD_EINVAL, "candidate %d outside the image", k);
        ent[k] = (u64)((uint32_t)y * (uint32_t)width + (uint32_t)x) | ((u64)(uint32_t)d << 32);       // sides = 0: cost only
    }
    const unsigned int ctr[16] = {(unsigned int)n};            // (counter[7] = the eval kernel's work counter)
    HIP_TRY(hipMemcpyAsync(c->xqueue.ptr, ent.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->xctr.ptr, ctr, sizeof(ctr), hipMemcpyHostToDevice, s));
    const double *d_prox = nullptr;
    if ((rc = get_prox64(*c, winSize, gammaP, s, &d_prox))) return rc;
    AswExactArgs x{};
    x.recL = (const PixRec *)c->recL.ptr; x.recR = (const PixRec *)c->recR.ptr;
    x.labL = (const double *)c->xlabL.ptr; x.labR = (const double *)c->xlabR.ptr;
    x.prox = d_prox; x.q.entries = (u64 *)c->xqueue.ptr; x.q.counter = (unsigned int *)c->xctr.ptr; x.q.cap = (unsigned int)std::max(n, 1);
    x.ecost = (double *)c->xcost.ptr;
    x.H = height; x.W = width; x.win = winSize; x.pad = winSize / 2; x.row0 = 0; x.rows = height; x.gammaC = gammaC;
    hipLaunchKernelGGL(asw_exact_eval_kernel, dim3(256), dim3(64 * EXACT_WAVES), 0, s, x);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(costs, c->xcost.ptr, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (lab1) HIP_TRY(hipMemcpyAsync(lab1, c->xlabL.ptr, npix * 24, hipMemcpyDeviceToHost, s));
    if (lab2) HIP_TRY(hipMemcpyAsync(lab2, c->xlabR.ptr, npix * 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SSAMD_OK;
}

int ssamd_gsw_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                     int winSize, int maxDisparity, int minDisparity, int gamma, float fMax, int iterations, int bins,
                     int16_t *d_disparity, void *stream)
{
    (void)bins;                       // never read by the reference either (_passive.cpp:410)
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    return gsw_device_impl(*c, d_img1, d_img2, height, width, out_row0, out_rows, winSize, maxDisparity, minDisparity,
                           gamma, fMax, iterations, d_disparity, (hipStream_t)stream);
}

// ssamd_gsw_device on TWO row ranges (round 6: the border bands of a row strip whose interior rows ran while the halo was in flight,
// strips.py).  GSW workgroups are small (two 8-wave groups per CU, strips of 2-8 rows) and a band is winSize/2 = 5 rows at the class
// default: each band is an ordinary call on its rows -- pack, both passes, left-right check -- written at its place in d_disparity.
int ssamd_gsw_device_rows2(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                           int skip_row0, int skip_rows, int winSize, int maxDisparity, int minDisparity, int gamma, float fMax,
                           int iterations, int bins, int16_t *d_disparity, void *stream)
{
    (void)bins;
    if (!d_img1 || !d_img2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    if (skip_rows < 0 || (skip_rows > 0 && (skip_row0 < out_row0 || skip_row0 + skip_rows > out_row0 + out_rows)))
        return fail(SSAMD_EINVAL, "the skipped rows [%d,%d) must lie inside the output rows [%d,%d)", skip_row0, skip_row0 + skip_rows,
                    out_row0, out_row0 + out_rows);
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    if (skip_rows == 0)
        return gsw_device_impl(*c, d_img1, d_img2, height, width, out_row0, out_rows, winSize, maxDisparity, minDisparity,
                               gamma, fMax, iterations, d_disparity, (hipStream_t)stream);
    if ((rc = check_common(height, width, winSize, minDisparity, maxDisparity, out_row0, out_rows))) return rc;
    const int top = skip_row0 - out_row0, bot0 = skip_row0 + skip_rows, bot = out_row0 + out_rows - bot0;
    if (top > 0 && (rc = gsw_device_impl(*c, d_img1, d_img2, height, width, out_row0, top, winSize, maxDisparity, minDisparity,
                                         gamma, fMax, iterations, d_disparity, (hipStream_t)stream)))
        return rc;
    if (bot > 0 && (rc = gsw_device_impl(*c, d_img1, d_img2, height, width, bot0, bot, winSize, maxDisparity, minDisparity,
                                         gamma, fMax, iterations, d_disparity + (size_t)(bot0 - out_row0) * width, (hipStream_t)stream)))
        return rc;
    return SSAMD_OK;
}

int ssamd_gsw_rectified_device(const uint8_t *d_raw1, const uint8_t *d_raw2, int src_height, int src_width,
                               const float *d_mapx1, const float *d_mapy1, const float *d_mapx2, const float *d_mapy2,
                               int height, int width, int interpolation, int winSize, int maxDisparity, int minDisparity,
                               int gamma, float fMax, int iterations, int bins, int16_t *d_disparity, void *stream)
{
    (void)bins;
    if (!d_raw1 || !d_raw2 || !d_mapx1 || !d_mapy1 || !d_mapx2 || !d_mapy2 || !d_disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    if (src_height <= 0 || src_width <= 0) return fail(SSAMD_EINVAL, "Wrong image dimensions!");
    if (interpolation != 0 && interpolation != 1) return fail(SSAMD_EINVAL, "only INTER_NEAREST (0) and INTER_LINEAR (1) are supported");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    RemapSrc rm;
    rm.src1 = d_raw1; rm.src2 = d_raw2; rm.mapx1 = d_mapx1; rm.mapy1 = d_mapy1; rm.mapx2 = d_mapx2; rm.mapy2 = d_mapy2;
    rm.Hs = src_height; rm.Ws = src_width; rm.nearest = interpolation == 0 ? 1 : 0;
    return gsw_device_impl(*c, nullptr, nullptr, height, width, 0, height, winSize, maxDisparity, minDisparity, gamma, fMax, iterations,
                           d_disparity, (hipStream_t)stream, &rm);
}

int ssamd_gsw(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
              int minDisparity, int gamma, float fMax, int iterations, int bins, int16_t *disparity, int device)
{
    (void)bins;
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    return gsw_host_rows(gsw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gamma, fMax, iterations,
                                 disparity), device);
}

int ssamd_gsw_multi(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                    int minDisparity, int gamma, float fMax, int iterations, int bins, int16_t *disparity,
                    const int *devices, int n_devices)
{
    (void)bins;
    if (!img1 || !img2 || !disparity) return fail(SSAMD_EINVAL, "NULL buffer");
    return run_strips(gsw_job(img1, img2, height, width, winSize, maxDisparity, minDisparity, gamma, fMax, iterations,
                              disparity), devices, n_devices, gsw_host_rows);
}

int ssamd_remap_bgr_device(const uint8_t *d_src, int src_h, int src_w, const float *d_mapx, const float *d_mapy,
                           int dst_h, int dst_w, int interpolation, uint8_t *d_dst, void *stream)
{
    if (!d_src || !d_mapx || !d_mapy || !d_dst) return fail(SSAMD_EINVAL, "NULL buffer");
    if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return fail(SSAMD_EINVAL, "Wrong image dimensions!");
    if (interpolation != 0 && interpolation != 1) return fail(SSAMD_EINVAL, "only INTER_NEAREST (0) and INTER_LINEAR (1) are supported");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long long npix = (long long)dst_h * dst_w;
    // (a thread owns four output pixels; the map and output pointers are 16- / 4-byte aligned: device allocations)
    if (((uintptr_t)d_mapx | (uintptr_t)d_mapy) & 15 || ((uintptr_t)d_dst & 3)) return fail(SSAMD_EINVAL, "maps must be 16-byte and the output 4-byte aligned");
    const int blocks = (int)std::min<long long>((npix / 4 + 255) / 256 + 1, 256 * 16);
    Timed t(*c, s, SSAMD_K_REMAP);
    hipLaunchKernelGGL(remap_bgr_kernel, dim3(blocks), dim3(256), 0, s, d_src, src_h, src_w, d_mapx, d_mapy, d_dst, npix,
                       interpolation == 0 ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return SSAMD_OK;
}

int ssamd_reproject_device(const int16_t *d_disparity, int h, int w, const double *Q, float *d_points, void *stream)
{
    if (!d_disparity || !Q || !d_points) return fail(SSAMD_EINVAL, "NULL buffer");
    if (h <= 0 || w <= 0) return fail(SSAMD_EINVAL, "Wrong image dimensions!");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Mat4 q;
    for (int k = 0; k < 16; ++k) q.m[k] = Q[k];
    if (h > 65535) return fail(SSAMD_ELIMIT, "more than 65535 rows");
    if ((w & 3) == 0 && (((uintptr_t)d_disparity & 7) || ((uintptr_t)d_points & 15)))
        return fail(SSAMD_EINVAL, "disparity / point buffers must be 8- / 16-byte aligned");
    const int per_row = (w & 3) == 0 ? w / 4 : w;                 // threads a row needs
    Timed t(*c, s, SSAMD_K_REPROJECT);
    hipLaunchKernelGGL(reproject_kernel, dim3((per_row + 255) / 256, h), dim3(256), 0, s, d_disparity, d_points, h, w, q);
    HIP_TRY(hipGetLastError());
    return SSAMD_OK;
}

namespace {
__global__ void debug_libm_kernel(int which, int n, const void *in, void *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (which == 0) reinterpret_cast<double *>(out)[i] = glibc_exp(reinterpret_cast<const double *>(in)[i]);
    else if (which == 1) reinterpret_cast<float *>(out)[i] = glibc_powf_pos(reinterpret_cast<const float *>(in)[i], (float)(1 / 3.0));
    else if (which == 2) reinterpret_cast<double *>(out)[i] = ssamd::exact_sqrt(reinterpret_cast<const double *>(in)[i]);
    else reinterpret_cast<double *>(out)[i] = reinterpret_cast<const double *>(in)[i] / 0.7 + reinterpret_cast<const double *>(in)[i] / 5.0;
}
}  // namespace

// Verification / diagnosis: the queues of the LAST exact call on the current device.  which = 0: the final queue (entries
// re-evaluated in fp64), 1: the raw queue of a merging call with its cost images.  Copies up to max_n entries; *n = entries appended.
int ssamd_debug_exact_queue(int which, long long max_n, unsigned long long *entries, unsigned int *keys, long long *n)
{
    if (!n || max_n < 0 || (max_n > 0 && !entries)) return fail(SSAMD_EINVAL, "bad argument");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    *n = 0;
    if (!c->xflags.ptr || c->exact_calls == 0) return SSAMD_OK;
    HIP_TRY(hipDeviceSynchronize());
    unsigned int ctr[5] = {0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(ctr, c->xflags.ptr, sizeof(ctr), hipMemcpyDeviceToHost));
    const unsigned int cnt = which ? ctr[4] : ctr[0], cap = which ? c->xrawcap : c->xcap;
    *n = cnt;
    const size_t m = (size_t)std::min<long long>(std::min<unsigned int>(cnt, cap), max_n);
    if (m == 0) return SSAMD_OK;
    const void *src = which ? c->xraw.ptr : c->xqueue.ptr;
    if (!src) return SSAMD_OK;
    HIP_TRY(hipMemcpy(entries, src, m * 8, hipMemcpyDeviceToHost));
    if (which && keys) HIP_TRY(hipMemcpy(keys, (const char *)src + (size_t)c->xrawcap * 8, m * 4, hipMemcpyDeviceToHost));
    return SSAMD_OK;
}

int ssamd_debug_libm(int which, int n, const void *in, void *out)
{
    if (!in || !out || n < 0 || which < 0 || which > 3)
        return fail(SSAMD_EINVAL, "ssamd_debug_libm: which = 0 (exp, doubles), 1 (powf(x, 1/3), floats), 2 (sqrt, doubles), 3 (x / 0.7 + x / 5.0, doubles)");
    if (n == 0) return SSAMD_OK;
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    const size_t bytes = (size_t)n * (which == 1 ? 4 : 8);
    if ((rc = c->lab.reserve(2 * bytes))) return rc;
    char *const d_in = (char *)c->lab.ptr, *const d_out = d_in + bytes;
    HIP_TRY(hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(debug_libm_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, which, n, (const void *)d_in, (void *)d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSAMD_OK;
}

int ssamd_debug_gsw_sqrt(int n, float *out)
{
    int what = 0;
    if (n < 0) { what = 1; n = -n; }       // n < 0: the bare v_sqrt_f32 over 0 .. -n - 1 (measurement of why it is not used)
    if (!out || n <= 0 || n > GSW_TAB_SIZE) return fail(SSAMD_EINVAL, "bad argument");
    CtxLock c;
    int rc = get_ctx(-1, c);
    if (rc) return rc;
    if ((rc = c->lab.reserve((size_t)n * 4))) return rc;
    hipStream_t s = c->stream;
    ScratchOrder order(*c, s);
    hipLaunchKernelGGL(gsw_sqrt_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (float *)c->lab.ptr, n, what);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->lab.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SSAMD_OK;
}

int ssamd_profile_enable(int on)
{
    for (auto &c : g_ctx) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.prof.on = on != 0;
    }
    return SSAMD_OK;
}

int ssamd_profile_reset(void)
{
    DeviceGuard guard;
    for (auto &c : g_ctx) {
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.dev < 0) continue;
        (void)hipSetDevice(c.dev);
        c.prof.drain();
        std::fill(c.prof.ms, c.prof.ms + SSAMD_K_COUNT, 0.0);
        std::fill(c.prof.n, c.prof.n + SSAMD_K_COUNT, 0LL);
    }
    return SSAMD_OK;
}

int ssamd_profile_read(double *ms, long long *launches)
{
    DeviceGuard guard;
    for (int k = 0; k < SSAMD_K_COUNT; ++k) { if (ms) ms[k] = 0; if (launches) launches[k] = 0; }
    for (auto &c : g_ctx) {
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.dev < 0) continue;
        (void)hipSetDevice(c.dev);
        c.prof.drain();
        for (int k = 0; k < SSAMD_K_COUNT; ++k) { if (ms) ms[k] += c.prof.ms[k]; if (launches) launches[k] += c.prof.n[k]; }
    }
    return SSAMD_OK;
}

}  // extern "C"

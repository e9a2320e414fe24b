/*
 * ssamd.h -- C ABI of libssamd.so, the MI355X (gfx950) implementation of the
 * SimpleStereo passive-matching hot path (Adaptive / Geodesic Support-Weight
 * stereo).  Plain pointers and sizes only: this is the drop-in boundary a
 * maintainer of the reference would bind instead of the CPython extension
 * `simplestereo._passive` (see INTEGRATION.md for the ctypes stub).
 *
 * Reference interface each entry point replaces (paths under the reference repo):
 *
 *   ssamd_asw          <->  _passive.computeASW   simplestereo/_passive.cpp:293-400
 *                           called from StereoASW.compute, passive.py:88-90
 *   ssamd_gsw          <->  _passive.computeGSW   simplestereo/_passive.cpp:703-774
 *                           called from StereoGSW.compute, passive.py:153-156
 *   ssamd_*_device     same operators on buffers already resident in HBM (no
 *                      reference counterpart: the reference has no device).
 *
 * Conventions
 *   - images: uint8, C-contiguous [height][width][3], channel order B,G,R
 *     (what cv2.imread returns; _passive.cpp:333-334 assumes the same).
 *   - disparity: int16, C-contiguous [rows][width], caller-allocated.
 *   - every function returns 0 on success or a negative SSAMD_E* code; a
 *     human-readable message for the calling thread is at ssamd_last_error().
 *   - the library never keeps a caller pointer after returning.
 *   - calls are serialised PER DEVICE by an internal mutex (ctypes releases the
 *     GIL during the call, the reference holds it: blocking semantics are kept);
 *     threads that target different devices run concurrently.
 *   - an entry point leaves the calling thread's current HIP device as it found it.
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     fails with SSAMD_ENODEVICE.
 */
#ifndef SSAMD_H
#define SSAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSAMD_ABI_VERSION 5      /* 5: ssamd_asw_exact_device_rows2 / _rectified_device (round 6: near-ties selected inside the aggregation kernels); 4: ssamd_asw_exact* (round 5); 3: GSW autotuning; 2: ssamd_set_option (round 3) + the multi-device and verification entry points added in round 2 */

#define SSAMD_OK 0
#define SSAMD_EINVAL (-1)     /* bad argument (message tells which)            */
#define SSAMD_ENODEVICE (-2)  /* no usable HIP device                          */
#define SSAMD_EHIP (-3)       /* a HIP runtime call failed                     */
#define SSAMD_ENOMEM (-4)     /* device or host allocation failed              */
#define SSAMD_ELIMIT (-5)     /* parameters exceed what the kernels support    */

int ssamd_abi_version(void);
const char *ssamd_last_error(void);
/* number of visible HIP devices (0 if none / runtime unavailable) */
int ssamd_device_count(void);

/* ---- host-buffer operators (H2D copy, kernels, D2H copy, synchronous) ------ */

/* Adaptive Support-Weight matching; argument meaning as _passive.computeASW
 * ("O!O!iiidd|p", _passive.cpp:301).  device: HIP device ordinal, or -1 for the
 * current device. */
int ssamd_asw(const uint8_t *img1, const uint8_t *img2, int height, int width,
              int winSize, int maxDisparity, int minDisparity,
              double gammaC, double gammaP, int consistent,
              int16_t *disparity, int device);

/* Geodesic Support-Weight matching; argument meaning as _passive.computeGSW
 * ("O!O!iiiifii", _passive.cpp:709).  `bins` is accepted and unused, like the
 * reference (workerGSW never reads it). */
int ssamd_gsw(const uint8_t *img1, const uint8_t *img2, int height, int width,
              int winSize, int maxDisparity, int minDisparity,
              int gamma, float fMax, int iterations, int bins,
              int16_t *disparity, int device);

/* ---- the same operators over several GPUs of one process ------------------- */
/* The frame is cut into n_devices contiguous row strips (heights differ by at most one row; empty when there are
 * more devices than rows); strip k, with its winSize/2 halo rows taken straight from the host arrays, is uploaded to
 * devices[k], matched there and copied back into rows of `disparity` -- one host thread per device, so copies and
 * kernels of all devices overlap.  Rows are the reference's independent jobs (_passive.cpp:372-374) and the
 * left-right check / occlusion filling are row-local (251-285), so the result is bit-identical to ssamd_asw /
 * ssamd_gsw on one device.  devices: distinct non-negative HIP ordinals.  No reference counterpart (the reference
 * has no device); SURVEY.md section 5 `devices=` extension of StereoASW / StereoGSW.compute. */
int ssamd_asw_multi(const uint8_t *img1, const uint8_t *img2, int height, int width,
                    int winSize, int maxDisparity, int minDisparity,
                    double gammaC, double gammaP, int consistent,
                    int16_t *disparity, const int *devices, int n_devices);
int ssamd_gsw_multi(const uint8_t *img1, const uint8_t *img2, int height, int width,
                    int winSize, int maxDisparity, int minDisparity,
                    int gamma, float fMax, int iterations, int bins,
                    int16_t *disparity, const int *devices, int n_devices);

/* ---- device-buffer operators (asynchronous on `stream`) -------------------- */
/* d_img1/d_img2: device pointers to a [height][width][3] sub-image (for a row
 * strip: the strip plus its winSize/2 halo rows).  Rows [out_row0,
 * out_row0+out_rows) of that sub-image are matched and written to d_disparity
 * ([out_rows][width]).  Image borders are the sub-image borders, so a strip that
 * carries its full halo reproduces the whole-image result exactly.
 * stream: a hipStream_t (NULL = the null stream).  The caller synchronises. */
int ssamd_asw_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width,
                     int out_row0, int out_rows,
                     int winSize, int maxDisparity, int minDisparity,
                     double gammaC, double gammaP, int consistent,
                     int16_t *d_disparity, void *stream);

/* ssamd_asw_device on TWO row ranges in one launch: rows [out_row0, skip_row0) and [skip_row0 + skip_rows, out_row0 + out_rows)
 * are matched, the rows in between are left untouched in d_disparity ([out_rows][width], laid out for the whole range).  For a row
 * strip of a frame cut across GPUs (simplestereo_amd/strips.py): the interior rows, whose windows stay inside the rows the rank owns,
 * run as an ordinary ssamd_asw_device call while the halo rows are still in flight over RCCL; the two border bands then take ONE
 * launch instead of two part-filled ones.  Rows are independent jobs in the reference (_passive.cpp:372-374), so the strip's map does
 * not depend on the cut.  skip_rows = 0: identical to ssamd_asw_device. */
int ssamd_asw_device_rows2(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width,
                           int out_row0, int out_rows, int skip_row0, int skip_rows,
                           int winSize, int maxDisparity, int minDisparity,
                           double gammaC, double gammaP, int consistent,
                           int16_t *d_disparity, void *stream);

int ssamd_gsw_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width,
                     int out_row0, int out_rows,
                     int winSize, int maxDisparity, int minDisparity,
                     int gamma, float fMax, int iterations, int bins,
                     int16_t *d_disparity, void *stream);

/* ssamd_gsw_device on two row ranges (same meaning as ssamd_asw_device_rows2; each band is matched by a call of its own:
 * GSW's workgroups are small and a band is winSize/2 rows).  Reference anchor: row-local jobs and row-local left-right check,
 * _passive.cpp:754-771, 661-696. */
int ssamd_gsw_device_rows2(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width,
                           int out_row0, int out_rows, int skip_row0, int skip_rows,
                           int winSize, int maxDisparity, int minDisparity,
                           int gamma, float fMax, int iterations, int bins,
                           int16_t *d_disparity, void *stream);

/* ---- ASW with the reference's fp64 argmin on near-ties ("exact" mode; what StereoASW runs by default since round 6) -----
 * The reference aggregates in double (_passive.cpp:23, 56-95); ssamd_asw* accumulate in fp32 and may pick the other one of
 * two candidates whose costs agree to ~1e-6 relative (a fraction of a percent of the pixels at worst).  These entry points
 * run the same kernels, whose epilogue queues every candidate that is a near-tie of its pixel's winner (within 1.5e-5 relative at
 * gammaC = 5 / winSize <= 35, wider for smaller gammaC and larger windows; on the saturated side within the reference's own fp64
 * rounding noise, ~6 win^2 2^-53 40 absolute), re-evaluate the queue in fp64 -- the reference's expression and summation order
 * (_passive.cpp:37-50, 57-88), fp64 CIELab (colorconversion.hpp:67-69), no contraction -- and redo those argmins (first minimum
 * wins, :90-93 / 243-246); both the left- and the right-referenced pass with `consistent`.  Scratch: O(H*W) (fp64 Lab images
 * 48 B / pixel, queue and slots 40 B / pixel; round 5 needed H*W*nD*4 bytes).  The weights are the reference's to the bit (glibc's
 * exp and powf restated for the device, csrc/glibc_math.hip.h; IEEE sqrt and division), so candidates one ulp apart resolve as in
 * the reference too: on the goldens, the whole bench frames and 77 000 random frames (rounds 5 and 6) the map IS the reference's.  Among exactly
 * equal fp64 costs the smallest index wins, as in ssamd_asw.  Candidates whose fp32 cost is exactly 0 are not queued: pixels with two
 * or more of them (the black margins of rectified frames) are settled by an integer test -- every in-image tap of the winner's window
 * has TAD = 0 => the reference's cost is exactly 0.0 and the winner is its first minimum -- and re-evaluated in full only where that
 * test fails.  A queue overflow (ssamd_counter "exact_overflow") keeps the fp32
 * map.  Same arguments and buffers as ssamd_asw / ssamd_asw_device / ssamd_asw_device_rows2 / ssamd_asw_rectified_device. */
int ssamd_asw_exact(const uint8_t *img1, const uint8_t *img2, int height, int width,
                    int winSize, int maxDisparity, int minDisparity,
                    double gammaC, double gammaP, int consistent,
                    int16_t *disparity, int device);
int ssamd_asw_exact_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width,
                           int out_row0, int out_rows,
                           int winSize, int maxDisparity, int minDisparity,
                           double gammaC, double gammaP, int consistent,
                           int16_t *d_disparity, void *stream);
int ssamd_asw_exact_device_rows2(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width,
                                 int out_row0, int out_rows, int skip_row0, int skip_rows,
                                 int winSize, int maxDisparity, int minDisparity,
                                 double gammaC, double gammaP, int consistent,
                                 int16_t *d_disparity, void *stream);
/* ssamd_asw_multi with the tie-break pass: one row strip per listed GPU, each strip tie-broken on its own device. */
int ssamd_asw_exact_multi(const uint8_t *img1, const uint8_t *img2, int height, int width,
                          int winSize, int maxDisparity, int minDisparity,
                          double gammaC, double gammaP, int consistent,
                          int16_t *disparity, const int *devices, int n_devices);

/* ---- "alternate pixel" ASW -------------------------------------
 * The faster variant the reference only sketches in a docstring todo (passive.py:43-46: "compute
 * disparity map on every other pixel with the traditional algorithm, then fill the remaining
 * pixels using left-right disparity boundaries"); opt-in, never the default.  Even image rows are
 * matched exactly; a pixel of an odd row searches only the disparities between the results of the
 * pixels above and below it (copied when they agree), with the exact ASW cost.  With `consistent`
 * the even rows also get the right-referenced pass, the left-right check and the occlusion filling
 * of ssamd_asw before the odd rows are derived from them.  Whole images only.  Same buffers and
 * error codes as ssamd_asw / ssamd_asw_device. */
int ssamd_asw_alternate(const uint8_t *img1, const uint8_t *img2, int height, int width,
                        int winSize, int maxDisparity, int minDisparity,
                        double gammaC, double gammaP, int consistent,
                        int16_t *disparity, int device);
int ssamd_asw_alternate_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width,
                               int winSize, int maxDisparity, int minDisparity,
                               double gammaC, double gammaP, int consistent,
                               int16_t *d_disparity, void *stream);

/* The same mode on a row range of a (sub-)image (row strips of a frame cut across GPUs or processes): rows whose index
 * in the WHOLE image is even are matched exactly, the odd ones filled from their two exact neighbours.  row_parity =
 * parity (0 / 1) of the sub-image's row 0 in the whole image.  A range that starts or ends with an odd row needs the
 * exact row just outside it: the sub-image must carry winSize/2 + 1 halo rows (fewer only at the borders of the whole
 * image).  d_disparity is int16 [out_rows][width]. */
int ssamd_asw_alternate_rows_device(const uint8_t *d_img1, const uint8_t *d_img2, int height, int width, int out_row0, int out_rows,
                                    int row_parity, int winSize, int maxDisparity, int minDisparity, double gammaC, double gammaP,
                                    int consistent, int16_t *d_disparity, void *stream);

/* ssamd_asw_multi for the alternate-rows mode: one row strip per listed GPU (halo of winSize/2 + 1 rows). */
int ssamd_asw_alternate_multi(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, int maxDisparity,
                              int minDisparity, double gammaC, double gammaP, int consistent, int16_t *disparity,
                              const int *devices, int n_devices);

/* ---- the steps either side of the matchers, on device --------- */

/* RectifiedStereoRig.rectifyImages (reference _rigs.py:543-567 = cv2.remap with constant
 * border): d_dst[y][x] = bilinear(d_src, d_mapx[y][x], d_mapy[y][x]).  d_src is uint8
 * [src_h][src_w][3]; maps are float32 [dst_h][dst_w] (cv2.initUndistortRectifyMap layout,
 * built once per rig on the host); interpolation 0 = nearest, 1 = linear.  The maps must be 16-byte and d_dst 4-byte
 * aligned (any device allocation is): a thread owns four consecutive output pixels. */
int ssamd_remap_bgr_device(const uint8_t *d_src, int src_h, int src_w, const float *d_mapx, const float *d_mapy,
                           int dst_h, int dst_w, int interpolation, uint8_t *d_dst, void *stream);

/* rectifyImages + StereoASW.compute in one call (the reference's own pipeline, examples/009 StereoMatchingASW.py:20-39:
 * _rigs.py:543-567 feeding passive.py:88): the two RAW frames d_raw1 / d_raw2 (uint8 [src_height][src_width][3], device
 * memory) are remapped through the rig's float32 maps [height][width] (device memory, 16-byte aligned) straight into the
 * matcher's pixel records -- rectification and CIELab conversion are ONE launch and the rectified BGR frames never exist in
 * HBM.  Bit-identical to ssamd_remap_bgr_device on each frame followed by ssamd_asw_device.  interpolation: 0 INTER_NEAREST,
 * 1 INTER_LINEAR.  d_disparity int16 [height][width]. */
int ssamd_asw_rectified_device(const uint8_t *d_raw1, const uint8_t *d_raw2, int src_height, int src_width,
                               const float *d_mapx1, const float *d_mapy1, const float *d_mapx2, const float *d_mapy2,
                               int height, int width, int interpolation, int winSize, int maxDisparity, int minDisparity,
                               double gammaC, double gammaP, int consistent, int16_t *d_disparity, void *stream);

/* ... with the fp64 tie-break pass (ssamd_asw_exact*): the records carry the remapped bytes, which is all the pass reads */
int ssamd_asw_exact_rectified_device(const uint8_t *d_raw1, const uint8_t *d_raw2, int src_height, int src_width,
                                     const float *d_mapx1, const float *d_mapy1, const float *d_mapx2, const float *d_mapy2,
                                     int height, int width, int interpolation, int winSize, int maxDisparity, int minDisparity,
                                     double gammaC, double gammaP, int consistent, int16_t *d_disparity, void *stream);

/* The same for StereoGSW (_rigs.py:543-567 feeding passive.py:153): raw frames through the rig's maps straight into the
 * matcher's packed pixels, one launch for both images. */
int ssamd_gsw_rectified_device(const uint8_t *d_raw1, const uint8_t *d_raw2, int src_height, int src_width,
                               const float *d_mapx1, const float *d_mapy1, const float *d_mapx2, const float *d_mapy2,
                               int height, int width, int interpolation, int winSize, int maxDisparity, int minDisparity,
                               int gamma, float fMax, int iterations, int bins, int16_t *d_disparity, void *stream);

/* RectifiedStereoRig.get3DPoints (reference _rigs.py:569-628 = cv2.reprojectImageTo3D):
 * d_points float32 [h][w][3] from int16 disparities and the 4x4 matrix Q (16 doubles, row
 * major, HOST memory).  h <= 65535; when w is a multiple of 4 (four pixels per thread) d_disparity must be 8-byte and
 * d_points 16-byte aligned (any device allocation is). */
int ssamd_reproject_device(const int16_t *d_disparity, int h, int w, const double *Q, float *d_points, void *stream);

/* ---- verification / measurement helpers ------------------------------------ */

/* Raw left-referenced aggregated ASW costs, float32 [height][width][nD] with
 * nD = maxDisparity-minDisparity+1, NaN where the reference evaluates no
 * candidate (x-d < 0).  Host buffers, synchronous.  For tolerance tests. */
int ssamd_asw_costs(const uint8_t *img1, const uint8_t *img2, int height, int width,
                    int winSize, int maxDisparity, int minDisparity,
                    double gammaC, double gammaP, float *costs, int device);

/* The two raw winner-take-all results of the consistent mode BEFORE the left-right check and the occlusion
 * filling (_passive.cpp:188 and 248-250): left_disparity[y][x] = x - dBest of the left-referenced pass,
 * right_match[y][xr] = dBest of the right-referenced pass, i.e. the LEFT column the right pixel xr selects (0 when
 * its candidate loop is empty).  int16 [height][width] host buffers, synchronous.  Lets a test check the argmins
 * against the reference's fp64 costs and the finalisation kernel against a literal restatement separately. */
int ssamd_asw_argmins(const uint8_t *img1, const uint8_t *img2, int height, int width,
                      int winSize, int maxDisparity, int minDisparity,
                      double gammaC, double gammaP,
                      int16_t *left_disparity, int16_t *right_match, int device);

/* CIELab conversion used by ASW (replaces ColorConversion::ImageFromBGR2Lab,
 * headers/colorconversion.hpp:81-86); float32 [height][width][3]. Host buffers. */
int ssamd_bgr2lab(const uint8_t *img, int height, int width, float *lab, int device);

/* The device's restatements of the two libm functions the reference's ASW path calls (csrc/glibc_math.hip.h), evaluated on n HOST
 * values: which = 0: exp on doubles (reference _passive.cpp:47-50); which = 1: powf(x, (float)(1/3.0)) on floats
 * (headers/colorconversion.hpp:55-65).  Lets a test prove them equal to the host's libm bit for bit. */
int ssamd_debug_libm(int which, int n, const void *in, void *out);
/* The near-tie queues of the LAST ssamd_asw_exact* call on the current device: which = 0 the final queue (what was re-evaluated
 * in fp64), 1 the raw queue of a merging call (several disparity chunks / consistent) with the fp32 cost images in `keys`.
 * Entry = pix | d << 32 | sides << 48 (pix = (row - out_row0) * width + left column; sides 1 left-, 2 right-referenced).
 * Copies at most max_n entries; *n = entries appended (may exceed the queue's capacity: overflow). */
int ssamd_debug_exact_queue(int which, long long max_n, unsigned long long *entries, unsigned int *keys, long long *n);

/* The fp64 cost the tie-break pass computes for each of n candidates (yxd: n triples y, x, d of HOST ints) of a host image pair, and
 * optionally (non-NULL) the fp64 CIELab images [height][width][3] it reads: lets a test compare them with the oracle's bit for bit. */
int ssamd_debug_exact_costs(const uint8_t *img1, const uint8_t *img2, int height, int width, int winSize, double gammaC, double gammaP,
                            int n, const int *yxd, double *costs, double *lab1, double *lab2);

/* The GSW kernels' exact integer square root, evaluated on the device for s = 0 .. n-1 (n <= 195076)
 * into a HOST buffer: lets a test prove it equals (float)sqrt((double)s) over the whole domain. */
int ssamd_debug_gsw_sqrt(int n, float *out);

/* Kernel timing with HIP events recorded on the launch stream.  After
 * ssamd_profile_enable(1) every operator call brackets its kernels with events;
 * ssamd_profile_read() synchronises and returns accumulated milliseconds and
 * launch counts per kernel slot since the last ssamd_profile_reset(). */
#define SSAMD_K_LAB 0        /* bgr2lab records                                  */
#define SSAMD_K_ASW_AGG 1    /* ASW cost aggregation + WTA keys (dominant)       */
#define SSAMD_K_ASW_FIN 2    /* ASW key decode / LR check / occlusion fill       */
#define SSAMD_K_GSW_AGG 3    /* GSW weights + cost aggregation + WTA keys        */
#define SSAMD_K_GSW_FIN 4    /* GSW LR check / occlusion fill                    */
#define SSAMD_K_REMAP 5      /* rectification remap (bilinear)                    */
#define SSAMD_K_REPROJECT 6  /* disparity -> 3-D points                           */
#define SSAMD_K_ASW_ALT 7    /* alternate-rows mode: bounded search on the odd rows */
#define SSAMD_K_ASW_EXACT 8  /* fp64 tie-break pass of ssamd_asw_exact* (fp64 Lab, filter, winners, eval, resolve, patch) */
#define SSAMD_K_COUNT 9
int ssamd_profile_enable(int on);
int ssamd_profile_reset(void);
int ssamd_profile_read(double *ms /*[SSAMD_K_COUNT]*/, long long *launches /*[SSAMD_K_COUNT]*/);
const char *ssamd_kernel_name(int slot);

/* Experiment / test hooks.  The SSAMD_* environment variables (SSAMD_ASW_GEOM, SSAMD_ASW_PIPE,
 * SSAMD_ASW_WAVE, ... -- the same names are the option names here) are read once, when the library is loaded; this
 * call changes one of them afterwards (value NULL = back to what the environment had set when the library was loaded, or
 * unset).  No operator call reads the environment.  None of the
 * options changes a disparity map: that is what the tests using them assert.  SSAMD_EINVAL for an unknown name. */
int ssamd_set_option(const char *name, const char *value);

/* Diagnostic counters of one device's context (for tests and for finding performance cliffs; never needed in
 * production).  "evol_fallbacks": ASW calls that could not get the pre-computed TAD volume (device memory short) and
 * ran the phase-shifted kernel with in-kernel e tiles, or a workgroup kernel instead of the small-range wave kernel;
 * "evol_bytes": capacity of the volume buffer the context holds right now; "tail_splits": phase-shifted launches whose last
 * partial round of workgroups ran as half-width tiles; "exact_calls": ssamd_asw_exact* calls; of the LAST such call (these
 * synchronise the device): "exact_entries" candidates re-evaluated in fp64, "exact_flagged_left" / "exact_flagged_right"
 * pixels with near-ties, "exact_raw_entries" what the aggregation kernels of a merging call (several chunks / consistent) queued
 * against their tile-local winners before the filter, "exact_overflow" 1 when a candidate queue overflowed (the fp32 map was
 * kept); "static_tile_mismatch": SSAMD_ASW_STATIC=2 launches whose planned geometry did not equal a compile-time tile.
 * SSAMD_EINVAL for an unknown name. */
int ssamd_counter(int device, const char *name, long long *value);

/* Autotuning of the ASW launch geometry.  When it applies, the first ssamd_asw* call for a problem shape (width,
 * rows, winSize, number of disparities) times the best tile of every class of candidates on the call's own
 * buffers -- up to ten candidates, five launches each in round-robin order, once -- and later calls reuse the
 * fastest.  The disparity maps do not depend on the geometry.  on = 1: always; 0: never; -1 (the default, also
 * SSAMD_AUTOTUNE=-1): only for calls of at most 6e10 window taps (6-8 ms of kernel time: VGA / 720p frames,
 * small disparity ranges), where the trial launches cost at most ~0.4 s once and the cost model is least reliable.  The environment
 * variable SSAMD_AUTOTUNE=1 / 0 / -1 sets the initial mode.  Returns the previous mode.
 * Since ABI version 3 the same mode governs ssamd_gsw*: the first GSW call of a shape of at most 6e10 window taps (both passes)
 * times strips of 2 / 4 / 8 output rows whose thread groups fill whole waves, three launches each, and caches the fastest. */
int ssamd_autotune(int on);

/* Launch geometry chosen for an ASW problem (for DESIGN.md / bench reporting).
 * out[0..7] = tile_x, chunk_d, n_chunks, threads, lds_bytes, grid_x, grid_y, grid_z */
int ssamd_asw_geometry(int width, int rows, int winSize, int maxDisparity, int minDisparity, int *out);

/* Which form of the ASW aggregation kernel that geometry runs: out[0] = 1 for the phase-shifted kernel
 * (asw_aggregate_pipe_kernel: lanes along the disparity groups, pre-computed TAD volume), 0 for asw_aggregate_kernel;
 * out[1] = columns of the register tile (8 or 4); out[2] = tap columns per chunk (0: whole window rows);
 * out[3] = 1 when waves 0-3 build before they aggregate; out[4] = 8 or 4 when asw_aggregate_wave_kernel (small disparity
 * ranges: every wave builds the support weights of its own strip) runs with that many columns per lane, else 0.
 * out must hold 5 ints. */
int ssamd_asw_kernel_form(int width, int rows, int winSize, int maxDisparity, int minDisparity, int *out);

/* Same for a GSW problem; out[0..8] = tile_x, chunk_d, n_chunks, threads, lds_bytes, grid_x, grid_y,
 * grid_z, strip_rows (output rows per workgroup: 1 or 2) */
int ssamd_gsw_geometry(int width, int rows, int winSize, int maxDisparity, int minDisparity, int *out);

#ifdef __cplusplus
}
#endif
#endif /* SSAMD_H */

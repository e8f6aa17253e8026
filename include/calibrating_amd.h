/*
 * calibrating_amd.h -- C ABI of libcalibrating_amd.so: the MI355X (gfx950) stereo-depth hot path of
 * DIYer22/calibrating, `Stereo.get_depth(img1, img2)`.
 *
 * Every entry point replaces one native (cv2 / NumPy) call the reference makes on that path; the
 * file:line after "replaces" points into /root/reference/calibrating/.  All image / volume pointers
 * are DEVICE pointers unless the name ends in `_host`; `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Launches are asynchronous on `stream`; the compute entry points never synchronise the
 * device or a stream (camd_sgbm_status and the data-dependent count of camd_depth_to_point_cloud's caller excepted).
 * The init-time calls camd_sgbm_create / camd_sgbm_destroy / camd_sgbm_set_option(CAMD_OPT_PATH, CAMD_PATH_CONCURRENT)
 * allocate and free device memory (hipMalloc / hipFree synchronise implicitly) and are not for stream capture;
 * camd_sgbm_create fills the padding of its cost volume on a private stream and waits for that stream only.
 * Return value: 0 on success, a negative camd_status otherwise;
 * camd_last_error() then holds a message (thread-local).
 *
 * Handles are not thread-safe: use one handle per host thread / stream.
 */
#ifndef CALIBRATING_AMD_H
#define CALIBRATING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum camd_status {
    CAMD_OK = 0,
    CAMD_ERR_BAD_ARG = -1,     /* size/type mismatch (cv2 raises cv2.error there) */
    CAMD_ERR_UNSUPPORTED = -2, /* parameter outside what the kernels implement */
    CAMD_ERR_NO_DEVICE = -3,   /* no HIP device / not gfx950 */
    CAMD_ERR_HIP = -4,         /* a HIP runtime call failed */
    CAMD_ERR_NOMEM = -5
} camd_status;

const char* camd_last_error(void);
int camd_version(void);
/* 0 when a gfx950 device is usable by this process, CAMD_ERR_NO_DEVICE otherwise */
int camd_device_ok(void);

/* ---- SGBM ------------------------------------------------------------------------------------
 * replaces cv2.StereoSGBM_create(...) and .compute(left, right):
 *   stereo_matching.py:48-58 (create; field order = keyword order there, plus preFilterCap, mode)
 *   stereo_matching.py:63    (compute), stereo_matching.py:64 (getMinDisparity)                */
typedef struct camd_sgbm_params {
    int minDisparity;
    int numDisparities;
    int blockSize;
    int P1;
    int P2;
    int disp12MaxDiff;
    int preFilterCap;
    int uniquenessRatio;
    int speckleWindowSize;
    int speckleRange;
    int mode; /* CAMD_MODE_SGBM = 0 (5 paths; the reference's call), CAMD_MODE_HH = 1 (8 paths),
                 CAMD_MODE_SGBM_3WAY = 2 (four row stripes x three paths), CAMD_MODE_HH4 = 3 (4 paths) */
} camd_sgbm_params;
enum { CAMD_MODE_SGBM = 0, CAMD_MODE_HH = 1, CAMD_MODE_SGBM_3WAY = 2, CAMD_MODE_HH4 = 3 };

typedef struct camd_sgbm camd_sgbm;

/* Workspace (device) bytes a handle for these sizes allocates. 0 on bad arguments. */
size_t camd_sgbm_workspace_bytes(const camd_sgbm_params* p, int width, int height, int channels,
                                 int max_batch);
/* Allocates the per-pair workspace for `max_batch` pairs of width x height x channels (1 or 3) u8.
 * Parameters are normalised as cv2's computeDisparitySGBM does and no further: numDisparities is used AS GIVEN (218 stays
 * 218; only the internal volume is padded, to a multiple of 32 above 64), blockSize <= 0 -> 5 and otherwise only its
 * half blockSize / 2 is used (an even size acts as the next odd one, as in cv2), P1 <= 0 -> 2, P2 <= 0 -> 5, then
 * P2 = max(P2, P1 + 1), uniquenessRatio < 0 -> 10, disp12MaxDiff <= 0 -> 1, preFilterCap -> max(cap, 15) | 1.
 * Refused with CAMD_ERR_UNSUPPORTED and a message, never computed differently (cv2.StereoSGBM_create accepts all of
 * these; INTEGRATION.md section D says what each limit comes from): numDisparities > 512, blockSize > 15 (> 11 for
 * MODE_SGBM_3WAY), P2 > 24000, preFilterCap > 127, 0 < width - numDisparities <= blockSize / 2 (cv2's own result is
 * undefined there), MODE_SGBM_3WAY on images too low for its four stripes. */
int camd_sgbm_create(const camd_sgbm_params* p, int width, int height, int channels, int max_batch,
                     camd_sgbm** out);
int camd_sgbm_destroy(camd_sgbm* h);
/* left/right: u8 [batch][height][width*channels] with row pitch `pitch` bytes and `image_stride`
 * bytes between consecutive pairs; disp: int16 [batch][height][width] (disparity * 16, cv2's
 * fixed point) with `disp_pitch` / `disp_stride` bytes.  batch <= max_batch.                    */
int camd_sgbm_compute(camd_sgbm* h, const uint8_t* left, const uint8_t* right, size_t pitch,
                      size_t image_stride, int16_t* disp, size_t disp_pitch, size_t disp_stride,
                      int batch, void* stream);
/* geometry of the internal cost volume: x in cost coordinates [0,width1), d in [0,D), padded to Dp */
int camd_sgbm_query(const camd_sgbm* h, int* width1, int* D, int* Dp, int* minX1);
/* stage-wise parity hooks: copy the volume of pair `index` of the last compute, [height][width1][Dp]
 * int16, to dst (device).  which: 0 = matching cost C (incl. +P2), 1 = aggregated S,
 * 2 = raw disparity before median/speckle as int16 [height][width] */
int camd_sgbm_debug_copy(camd_sgbm* h, int which, int index, void* dst, void* stream);
/* options: CAMD_OPT_PATH selects the aggregation implementation (all bit-identical):
 *   CAMD_PATH_AUTO (default)  concurrent scans for calls of <= 4 pairs of 1080p/D=128-sized work (<= 8 for
 *                             MODE_HH), band passes above
 *   CAMD_PATH_SCAN            one line-scan launch per direction, sequential (generic fallback)
 *   CAMD_PATH_BAND            fused band-wavefront passes (throughput; D in (32, 256])
 *   CAMD_PATH_CONCURRENT      all directions at once into per-direction volumes (latency; <= 8 pairs per call)
 * MODE_SGBM_3WAY has no per-direction volumes: AUTO takes the line scans for little work per call (under ~2.5 pairs of
 * 1080p/D=128-sized work) and the band passes above; CAMD_PATH_BAND / CAMD_PATH_CONCURRENT force the band passes
 * (where D allows; the winners are decided inside the last pass for D % 8 == 0, by a separate kernel otherwise).
 * CAMD_OPT_KEEP_S 1 = the band path also stores the final S volume (for camd_sgbm_debug_copy(which = 1)). */
enum { CAMD_OPT_PATH = 0, CAMD_OPT_KEEP_S = 1, CAMD_OPT_COST = 2, CAMD_OPT_SATURATE = 3, CAMD_OPT_3WAY_SIMD_LANES = 4,
       CAMD_OPT_EXACT = 5 };  /* (6 = CAMD_OPT_PHASES: calibrating_amd_experimental.h) */
enum { CAMD_PATH_AUTO = 0, CAMD_PATH_SCAN = 1, CAMD_PATH_BAND = 2, CAMD_PATH_CONCURRENT = 3 };
/* CAMD_OPT_COST selects how the matching-cost volume C is built (bit-identical results):
 *   CAMD_COST_AUTO (default)  the fused kernel where it is instantiated (blockSize <= 11), else the split pair
 *   CAMD_COST_FUSED           k_cost: BT pixel cost + blockSize x blockSize box sum + P2 -> C, written once
 *   CAMD_COST_SPLIT           k_hsum (BT + horizontal sum) -> intermediate volume -> k_vsum (vertical sum) -> C */
enum { CAMD_COST_AUTO = 0, CAMD_COST_FUSED = 1, CAMD_COST_SPLIT = 2 };
/* CAMD_OPT_SATURATE: int16 overflow behaviour of the box-sum recurrences that build C:
 *   1 (default)  saturate like OpenCV's CV_SIMD build (v_int16 + / -), which is what cv2 wheels run
 *   0            wrap modulo 2^16 like OpenCV's scalar build
 * The two differ only if blockSize^2 * channels * (2*ftzero + 63) + P2 > 32767 AND the image content drives a
 * window sum past 32767 (e.g. blockSize >= 11 on RGB with a large preFilterCap).
 * CAMD_OPT_3WAY_SIMD_LANES (MODE_SGBM_3WAY only): 8 (default) = the winner-take-all tie rule of cv2's 8-lane SIMD builds
 * (per lane slot the last disparity attaining the slot minimum, then the smallest of those), 1 = the scalar build's
 * smallest disparity.  Only exact ties are affected.
 * CAMD_OPT_EXACT: what happens to a pair whose cost volume left the int16 regime of the aggregation kernels (a value
 * below P2, possible only after an int16 overflow of the box sums, i.e. only when blockSize^2 * channels *
 * (2*ftzero + 63) + P2 > 32767 and the images are adversarial):
 *   1 (default)  it is aggregated again in plain int arithmetic, exactly as OpenCV's scalar code does (slow, one set
 *                of per-direction volumes of workspace)
 *   0            it is refused: its disparities are written as invalid and camd_sgbm_status / the next
 *                camd_sgbm_compute return CAMD_ERR_HIP (also what happens when that workspace could not be allocated)
 * Either way a result that differs from OpenCV's is never handed back silently. */
int camd_sgbm_set_option(camd_sgbm* h, int option, int value);
/* synchronises `stream` and reports whether a device-side bounded wait of the last computes timed out, or a pair
 * was refused (CAMD_OPT_EXACT).  Without this call neither can pass unnoticed: the affected disparities are written as
 * invalid ((minDisparity - 1) * 16), and the next camd_sgbm_compute on the handle returns CAMD_ERR_HIP. */
int camd_sgbm_status(camd_sgbm* h, void* stream);
/* per-stage GPU time of the last compute, measured with hipEvents on `stream` (enable first).
 * stage names: camd_sgbm_stage_name(i), i in [0, camd_sgbm_num_stages()): "cost", "hsum", "vsum" (the split cost pair),
 * "scan" (aggregation; on the band path its first pass), "scan_last", "wta" (winner-take-all / LR check), "median",
 * "speckle" */
int camd_sgbm_set_profiling(camd_sgbm* h, int enable);
int camd_sgbm_num_stages(void);
const char* camd_sgbm_stage_name(int i);
int camd_sgbm_get_profile(camd_sgbm* h, float* ms_per_stage, int n);

/* replaces cv2.medianBlur(disp, disp, 3) and cv2.filterSpeckles inside StereoSGBM.compute; exported
 * for stage-wise parity.  src/dst int16 [batch][h][w] contiguous; dst != src.                      */
int camd_median3_s16(const int16_t* src, int16_t* dst, int w, int h, int batch, void* stream);
/* in place; labels_ws: device scratch of camd_speckle_workspace_bytes(w,h,batch) */
size_t camd_speckle_workspace_bytes(int w, int h, int batch);
int camd_filter_speckles_s16(int16_t* img, int w, int h, int new_val, int max_speckle_size,
                             int max_diff, void* labels_ws, int batch, void* stream);

/* ---- remaps ----------------------------------------------------------------------------------
 * replaces cv2.remap(img, mapx, mapy, interp) on u8 HWC with CV_32FC1 maps, BORDER_CONSTANT 0:
 *   stereo_camera.py:217-228 (INTER_LANCZOS4, both cameras)
 * x_shift implements stereo_camera.py:230-240 (translation_rectify_img): dst[:, x] takes the remap
 * result of column x - x_shift, vacated columns are 0.                                          */
enum { CAMD_INTER_NEAREST = 0, CAMD_INTER_LINEAR = 1, CAMD_INTER_LANCZOS4 = 4 };
int camd_remap_u8(const uint8_t* src, int sw, int sh, int cn, size_t src_pitch, size_t src_stride,
                  const float* mapx, const float* mapy, uint8_t* dst, int dw, int dh,
                  size_t dst_pitch, size_t dst_stride, int interp, int x_shift, int batch,
                  void* stream);
/* replaces cv2.undistort(img1, K, D) (stereo_camera.py:430-431): bilinear fixed-point remap through
 * the CV_16SC2 + CV_16UC1 maps cv2.undistort builds internally (camd_undistort_maps_host).       */
int camd_remap_fixed_bilinear_u8(const uint8_t* src, int sw, int sh, int cn, size_t src_pitch,
                                 size_t src_stride, const int16_t* mapxy, const uint16_t* mapa,
                                 uint8_t* dst, int dw, int dh, size_t dst_pitch, size_t dst_stride,
                                 int batch, void* stream);
/* host, init time: the stripe-wise fixed-point maps of cv2.undistort (2*w*h int16 + w*h uint16) */
int camd_undistort_maps_host(const double K[9], const double* dist, int ndist, int w, int h,
                             int16_t* mapxy_host, uint16_t* mapa_host);
/* ---- rig tables on the GPU (init time, or per batch when the rig / target size changes) -------------
 * replaces cv2.initUndistortRectifyMap(A, dist, R, Anew, (w, h), CV_32FC1):
 *   stereo_camera.py:159-165 (rectify maps of both cameras)   utils.py:184-191 (unrectify maps)
 * and, when valid_mask != NULL, valid_mask_from_remap (stereo_camera.py:167-176) against a src_w x src_h
 * source image, fused.  A, Anew: 3x3 row-major host doubles; R: 3x3 or NULL (identity); dist: up to 12
 * coefficients (k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4), host.  mapx / mapy: device float [h][w];
 * valid_mask: device u8 [h][w] or NULL.  Bit-identical to the host construction (float64 internally,
 * X/Y/W accumulated along each row like OpenCV's scalar loop).                                         */
int camd_init_undistort_rectify_map(const double A[9], const double* dist, int ndist, const double* R,
                                    const double Anew[9], int w, int h, float* mapx, float* mapy,
                                    uint8_t* valid_mask, int src_w, int src_h, void* stream);
/* device version of camd_undistort_maps_host: mapxy int16 [h][w][2], mapa uint16 [h][w] (device) */
int camd_undistort_maps(const double K[9], const double* dist, int ndist, int w, int h, int16_t* mapxy,
                        uint16_t* mapa, void* stream);
/* process-wide options.  CAMD_GOPT_LANCZOS_FIX_GROUP_LO: first index (3 or 4; default 4 = ksize/2) of the 2x2 tap
 * group of a Lanczos-4 table entry that takes the weight-sum correction -- SURVEY.md A.10 uncertainty U15, the same
 * switch the CPU oracle exposes (oracle_switches.lanczos_fix_group_lo), so that one flip moves both. */
enum { CAMD_GOPT_LANCZOS_FIX_GROUP_LO = 0 };
int camd_set_global_option(int option, int value);
/* host, init time: the 32x32-phase int16 weight tables cv2.remap uses (1024*64 / 1024*4 entries) */
int camd_lanczos4_table_host(int16_t* tab_host);
int camd_bilinear_table_host(int16_t* tab_host);

/* replaces boxx.resize -> cv2.resize(..., INTER_LINEAR) around the matcher when max(h, w) > cfg["max_size"]:
 *   stereo_matching.py:62 (u8 RGB pair, downsize)   stereo_matching.py:66 (float32 disparity, upsize)
 * src [batch][sh][sw][cn], dst [batch][dh][dw][cn], contiguous. */
int camd_resize_linear_u8(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh, int batch,
                          void* stream);
int camd_resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int batch,
                           void* stream);

/* ---- depth -----------------------------------------------------------------------------------
 * replaces stereo_matching.py:63-69 (int16 -> f32, clip, < minD*16 -> 0, /16, identity resize),
 * stereo_camera.py:510-512 (+= min_disparity, * rectify_valid_mask1) and
 * stereo_camera.py:408-413 (Stereo.disparity_to_depth) in one pass.
 * valid_mask: u8 [h][w] shared by the batch; disparity: f32, depth: f64 (NumPy >= 2 dtype).      */
int camd_disp_to_depth(const int16_t* disp16, const uint8_t* valid_mask, int w, int h,
                       int sgbm_min_disparity, int add_min_disparity, int translate,
                       double baseline_fx, double max_depth, float* disparity, double* depth,
                       int batch, void* stream);
/* The same for the matcher's downsizing branch (cfg["max_size"] < max(h, w), the reference's default 1000):
 * disp16 is the int16 disparity of the sw x sh DOWNSIZED pair; stereo_matching.py:63-69 in full -- float32, clip,
 * < minD*16 -> 0, /16, boxx.resize (cv2.resize INTER_LINEAR) back to w x h, * w / sw -- then stereo_camera.py:510-513
 * and :408-413 as above, one pass.                                                                                  */
int camd_disp16_resized_to_depth(const int16_t* disp16, int sw, int sh, const uint8_t* valid_mask, int w, int h,
                                 int sgbm_min_disparity, int add_min_disparity, int translate,
                                 double baseline_fx, double max_depth, float* disparity, double* depth,
                                 int batch, void* stream);
/* replaces utils.rotate_depth_by_remap (utils.py:192-199) as called by Stereo.unrectify_depth
 * (stereo_camera.py:415-428): z' = M20*(u*z) + M21*(v*z) + M22*z, then INTER_NEAREST remap.      */
int camd_unrectify_depth(const double* depth, int w, int h, const double M_row2_host[3],
                         const float* mapx, const float* mapy, double* out, int ow, int oh,
                         int batch, void* stream);

/* ---- depth post-ops (the step after get_depth in the reference's demos) ---------------------------
 * replaces utils.depth_to_point_cloud (utils.py:213-246): non-zero depths in row-major order of the sampling
 * grid (the depth image itself, or its cv2.resize(INTER_NEAREST) to round(size * rate) when rate != 1) ->
 * points [N][3] = Kinv * (u z, v z, z); uv (optional) [N][2] = the (u, v) of each point (return_xyzuv).
 * depth: f64 [h][w]; capacity = rows available in points/uv (grid_w * grid_h always suffices, see
 * camd_point_cloud_grid); *count (device u64) receives N; workspace: camd_point_cloud_workspace_bytes.   */
int camd_point_cloud_grid(int w, int h, double rate, int* grid_w, int* grid_h);
size_t camd_point_cloud_workspace_bytes(int w, int h, double rate);
int camd_depth_to_point_cloud(const double* depth, int w, int h, const double Kinv_host[9], double rate,
                              double* points, double* uv, size_t capacity, unsigned long long* count,
                              void* workspace, void* stream);
/* replaces utils.apply_T_to_point_cloud (utils.py:152-161): out = (T * [p, 1])[:3], T 4x4 row-major host */
int camd_apply_T_to_point_cloud(const double* points, size_t n, const double T_host[16], double* out,
                                void* stream);
/* replaces utils.point_cloud_to_depth / point_cloud_to_arr2d without values (utils.py:249-318): project with
 * K, round half-to-even to a pixel, nearest z wins (the reference sorts far-to-near and overwrites).
 * points: f64 rows of point_stride >= 3 doubles; keys_ws: device scratch of w*h*8 bytes; depth: f64 [h][w] */
int camd_point_cloud_to_depth(const double* points, size_t n, int point_stride, const double K_host[9], int w,
                              int h, double bg_value, double* depth, unsigned long long* keys_ws, void* stream);
/* replaces Cam.project_cam2_depth (camera.py:298-309) = the three calls above composed, as ONE scatter
 * pass without materialising the point cloud: depth2 f64 [h2][w2] of camera 2 -> depth1 f64 [h1][w1].      */
int camd_project_depth(const double* depth2, int w2, int h2, const double K2inv_host[9],
                       const double T_2in1_host[16], const double K1_host[9], double rate, int w1, int h1,
                       double* depth1, unsigned long long* keys_ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif

/*
 * oracle.h -- CPU restatement of the stereo-depth hot path of DIYer22/calibrating.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there
 * only as the checker / the reported CPU baseline.  The product path (calibrating_amd/) never
 * links, imports or executes it.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in a third-party dependency that is not
 * vendored in the reference tree: OpenCV (`opencv-contrib-python>=4.7.0.72`,
 * /root/reference/requirements.txt:2), modules calib3d (stereosgbm.cpp, undistort.dispatch.cpp)
 * and imgproc (imgwarp.cpp, median_blur).  cv2 is not importable in the authoring container and
 * the reference ships no golden vectors for this path, so this file restates OpenCV 4.x's
 * published algorithms from the reference's call sites:
 *   calibrating/stereo_matching.py:48-58,63   cv2.StereoSGBM_create(...).compute
 *   calibrating/stereo_camera.py:159-165      cv2.initUndistortRectifyMap
 *   calibrating/stereo_camera.py:217-228      cv2.remap(..., INTER_LANCZOS4)
 *   calibrating/stereo_camera.py:408-413      Stereo.disparity_to_depth
 *   calibrating/stereo_camera.py:430-431      cv2.undistort
 *   calibrating/utils.py:173-200              rotate_depth_by_remap (cv2.remap INTER_NEAREST)
 * It is pinned only by analytic known-answer tests and by an independent NumPy model in tests/.
 * The parts that restate the REFERENCE's own NumPy (depth_ref.c; oracle/pointcloud_ref.py) are pinned since round 5 by
 * outputs of the reference itself: tests/golden/reference_plumbing.npz, made by running /root/reference/calibrating
 * unmodified with these functions standing in for its cv2 entry points (tests/golden/make_reference_golden.py) --
 * which pins the reference's Python around the entry points, not OpenCV's arithmetic behind them.
 */
#ifndef CALIBRATING_ORACLE_H
#define CALIBRATING_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order as cv2.StereoSGBM_create's arguments (stereo_matching.py:48-58). */
typedef struct oracle_sgbm_params {
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
    int mode; /* 0 = MODE_SGBM (5 paths, reference's call), 1 = MODE_HH (8 paths), 2 = MODE_SGBM_3WAY, 3 = MODE_HH4 */
} oracle_sgbm_params;

/* switches for the points SURVEY.md Appendix A flags as uncertain (U-flags) */
typedef struct oracle_switches {
    int lanczos_fix_group_lo; /* U15: first index of the 2x2 tap group that takes the weight-sum
                                 correction; OpenCV source: ksize/2 (=4 for Lanczos4) */
    int bt_border_raw_tab0;   /* U11: raw-intensity planes also get tab[0] at columns 0, W-1 (1) */
    int cost_saturate;        /* U7: the box-sum recurrences that build C saturate to int16 like OpenCV's
                                 universal-intrinsic v_int16 + / - / * (1, what shipped SIMD builds run;
                                 default) or wrap like the scalar (CostType) casts (0).  Identical whenever
                                 blockSize^2 * cn * (2*ftzero + 63) + P2 <= 32767. */
    int way3_stripes;         /* U17: MODE_SGBM_3WAY row stripes; OpenCV fixes it at 4 "disregarding the number of
                                 threads to make the results fully reproducible" */
    int way3_simd_lanes;      /* U20: v_int16 lanes of the OpenCV build (8 = SSE / NEON, what x86-64 and aarch64 wheels
                                 use; 16 = AVX2 baseline): MODE_SGBM_3WAY's winner-take-all keeps, per lane slot, the
                                 LAST disparity that attains the slot minimum and then the smallest position among the
                                 slots holding the global minimum; <= 1 = the scalar build (smallest disparity wins) */
} oracle_switches;
void oracle_set_switches(const oracle_switches* s);
void oracle_get_switches(oracle_switches* s);

/* cv::StereoSGBM::compute : left/right u8 HxWxcn (row pitch `step` bytes) -> disp int16 HxW (x16).
 * Returns 0, or -1 on bad arguments. Stages: computeDisparitySGBM -> medianBlur(3) -> filterSpeckles. */
int oracle_sgbm_compute(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                        int width, int height, int cn, size_t step, int16_t* disp);
/* same, but stops after computeDisparitySGBM (no median, no speckle): for stage-wise parity */
int oracle_sgbm_raw(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                    int width, int height, int cn, size_t step, int16_t* disp);
/* batch of `n` independent pairs, `nthreads` OpenMP threads across pairs (CPU baseline leg) */
int oracle_sgbm_compute_batch(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                              int width, int height, int cn, int n, int nthreads, int16_t* disp);

/* stage internals exposed for stage-wise parity tests:
 * cost volume C[y][x][d] (int16, includes +P2), x in cost coordinates [0,width1), d in [0,D) */
int oracle_sgbm_cost_volume(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                            int width, int height, int cn, size_t step, int16_t* C);
/* aggregated S[y][x][d] after all passes of the selected mode */
int oracle_sgbm_aggregated(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                           int width, int height, int cn, size_t step, int16_t* S);

void oracle_median3_s16(const int16_t* src, int16_t* dst, int width, int height);
void oracle_filter_speckles_s16(int16_t* img, int width, int height, int newVal, int maxSpeckleSize,
                                int maxDiff);

/* cv2.remap on u8 HWC with CV_32FC1 map pair, BORDER_CONSTANT(0).
 * interp: 0 = INTER_NEAREST, 1 = INTER_LINEAR, 4 = INTER_LANCZOS4 */
int oracle_remap_u8(const uint8_t* src, int sw, int sh, int cn, const float* mapx, const float* mapy,
                    uint8_t* dst, int dw, int dh, int interp);
/* cv2.remap(float64 single-channel, INTER_NEAREST, BORDER_CONSTANT 0) */
void oracle_remap_nearest_f64(const double* src, int sw, int sh, const float* mapx,
                              const float* mapy, double* dst, int dw, int dh);
/* cv2.initUndistortRectifyMap(A, dist(ndist<=14 or NULL), R(or NULL), Anew, (w,h), CV_32FC1) */
void oracle_init_undistort_rectify_map(const double A[9], const double* dist, int ndist,
                                       const double* R, const double Anew[9], int w, int h,
                                       float* mapx, float* mapy);
/* cv2.undistort(img, K, D) on u8 HWC (stripe-wise CV_16SC2 maps + bilinear fixed point) */
void oracle_undistort_u8(const uint8_t* src, int w, int h, int cn, const double K[9],
                         const double* dist, int ndist, uint8_t* dst);
/* 32x32 phases x (8x8|2x2) int16 tables as cv::initInterTab2D builds them */
void oracle_lanczos4_itab(int16_t* tab /* 1024*64 */);
void oracle_bilinear_itab(int16_t* tab /* 1024*4 */);

/* cv2.resize(src, (dw, dh), interpolation=INTER_LINEAR) as boxx.resize calls it
 * (stereo_matching.py:62,66): u8 HWC (fixed point) and float32 single channel */
int oracle_resize_linear_u8(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh);
int oracle_resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh);

/* stereo_matching.py:63-69 (identity resize) + stereo_camera.py:510-513,408-413:
 * disp16 (int16 x16) -> disparity f32 (masked, +min_disparity) and rectified depth f64 */
void oracle_disp_to_depth(const int16_t* disp16, const uint8_t* valid_mask, int w, int h,
                          int sgbm_min_disparity, int add_min_disparity, int translate,
                          double baseline_fx, double max_depth, float* disparity, double* depth);
/* utils.py:192-199: z' = depth*(M20*u+M21*v+M22), then NN remap */
void oracle_unrectify_depth(const double* depth, int w, int h, const double Mrow2[3],
                            const float* mapx, const float* mapy, double* out, int ow, int oh);

#ifdef __cplusplus
}
#endif
#endif

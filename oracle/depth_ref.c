/*
 * depth_ref.c -- CPU restatement of the NumPy post-processing around the matcher.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Follows the reference's own Python line by line:
 *   /root/reference/calibrating/stereo_matching.py:63-69   int16 -> f32, clip(0), < minD*16 -> 0,
 *                                                          /16.0, (identity resize) * w / sw
 *   /root/reference/calibrating/stereo_camera.py:510-512   += min_disparity ; * rectify_valid_mask1
 *   /root/reference/calibrating/stereo_camera.py:408-413   depth = 1.0*baseline*fx/disparity (f64
 *                                                          under NumPy >= 2), > max_depth -> 0, < 0 -> 0
 *   /root/reference/calibrating/utils.py:192-199           z' = (R @ inv(K) @ [x*z, y*z, z])[2], NN remap
 */
#include "oracle.h"

void oracle_disp_to_depth(const int16_t* disp16, const uint8_t* valid_mask, int w, int h,
                          int sgbm_min_disparity, int add_min_disparity, int translate,
                          double baseline_fx, double max_depth, float* disparity, double* depth)
{
    const float thresh = (float)(sgbm_min_disparity * 16);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        float s = (float)disp16[i];
        if (s < 0.f) s = 0.f;                 /* .clip(0) */
        if (s < thresh) s = 0.f;              /* sdisparity[sdisparity < minDisparity*16] = 0 */
        float d = s / 16.0f;                  /* float32 / python float stays float32 */
        d = d * (float)w / (float)w;          /* boxx.resize identity, * img1.shape[1] / simg1.shape[1] */
        if (translate) d += (float)add_min_disparity; /* disparity += self.min_disparity */
        d = valid_mask[i] ? d : 0.f * d;      /* bool * float32 */
        disparity[i] = d;
        double z = baseline_fx / (double)d;   /* np.float64 scalar / float32 array -> float64 */
        if (z > max_depth) z = 0.;            /* inf from d == 0 is zeroed here */
        if (z < 0.) z = 0.;
        depth[i] = z;
    }
}

void oracle_unrectify_depth(const double* depth, int w, int h, const double M[3],
                            const float* mapx, const float* mapy, double* out, int ow, int oh)
{
    double* tmp = (double*)__builtin_malloc(sizeof(double) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double z = depth[(size_t)y * w + x];
            /* points = [xs, ys, 1] * depth ; new = M @ points ; row 2.  The matrix product is NumPy's matmul = a BLAS
             * dgemm, whose x86-64 kernels accumulate the k = 3 terms with fused multiply-adds:
             * fma(M2, z, fma(M1, y*z, M0*(x*z))) -- measured against the reference's own run (NumPy 2.2 / OpenBLAS,
             * tests/golden/reference_plumbing.npz: every unrectify_depth of the catalogue reproduces bit for bit this
             * way and in 3 of 4 pixels otherwise).  A BLAS without FMA kernels would round each product: 1 ulp apart. */
            tmp[(size_t)y * w + x] = __builtin_fma(M[2], z, __builtin_fma(M[1], (double)y * z, M[0] * ((double)x * z)));
        }
    oracle_remap_nearest_f64(tmp, w, h, mapx, mapy, out, ow, oh);
    __builtin_free(tmp);
}

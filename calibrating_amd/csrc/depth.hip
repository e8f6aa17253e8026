// depth.hip -- the NumPy post-processing around the matcher, fused into two gfx950 kernels.
//
// Replaces (file:line in /root/reference/calibrating/):
//   stereo_matching.py:63-69   compute(...).astype(float32).clip(0); [< minD*16] = 0; /16.0;
//                              boxx.resize (identity when max_size >= max(h,w)) * w / sw
//   stereo_camera.py:510-512   disparity += min_disparity ; disparity = rectify_valid_mask1 * disparity
//   stereo_camera.py:408-413   depth = 1.0*baseline*fx/disparity ; [> max_depth] = 0 ; [< 0] = 0
//   utils.py:192-199           rotate_depth_by_remap: z' = (R @ inv(K) @ [x*z, y*z, z])[2], then
//                              cv2.remap(..., INTER_NEAREST) through memoised maps
// depth is float64 (what NumPy >= 2 produces for np.float64 scalar / float32 array).
#include "common.hpp"

namespace camd {

// a thread owns two consecutive pixels: the int16 pair arrives as one dword, the float and double pairs leave as
// 8- and 16-byte stores
__global__ __launch_bounds__(256) void k_disp_to_depth(const int16_t* __restrict__ disp16,
                                                       const uint8_t* __restrict__ mask, int n,
                                                       float thresh, float addv, int translate, float wf,
                                                       double bf, double max_depth,
                                                       float* __restrict__ disparity, double* __restrict__ depth)
{
    const int i = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const bool two = i + 1 < n;
    const size_t o = (size_t)blockIdx.y * n + i;
    uint32_t both;
    if (two) __builtin_memcpy(&both, disp16 + o, 4);
    else both = (uint16_t)disp16[o];
    auto one = [&](int16_t raw, uint8_t m, float& dout, double& zout) {
        float s = (float)raw;
        s = s < 0.f ? 0.f : s;
        s = s < thresh ? 0.f : s;
        float d = s / 16.0f;
        d = __fdiv_rn(__fmul_rn(d, wf), wf);  // (d * w) / sw with sw == w, each op rounded like NumPy
        if (translate) d = __fadd_rn(d, addv);
        d = m ? d : __fmul_rn(0.f, d);
        dout = d;
        double z = __ddiv_rn(bf, (double)d);
        z = z > max_depth ? 0. : z;
        zout = z < 0. ? 0. : z;
    };
    float d0, d1 = 0.f;
    double z0, z1 = 0.;
    one((int16_t)(both & 0xffffu), mask[i], d0, z0);
    if (two) {
        one((int16_t)(both >> 16), mask[i + 1], d1, z1);
        const float dd[2] = {d0, d1};
        const double zz[2] = {z0, z1};
        __builtin_memcpy(disparity + o, dd, 8);
        __builtin_memcpy(depth + o, zz, 16);
    } else {
        disparity[o] = d0;
        depth[o] = z0;
    }
}

// The images of a batch share the rig's maps: blockIdx.z owns `zb` consecutive images, so the map pair, the rounding
// and the bounds test of a destination pixel are done once for all of them (8 of the 24 bytes per pixel and image).
__global__ __launch_bounds__(256) void k_unrectify(const double* __restrict__ depth, int w, int h, double m0,
                                                   double m1, double m2, const float* __restrict__ mapx,
                                                   const float* __restrict__ mapy, double* __restrict__ out,
                                                   int ow, int oh, int batch, int zb)
{
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y;
    if (x >= ow) return;
    const int z0 = blockIdx.z * zb, nz = min(zb, batch - z0);
    size_t mi = (size_t)y * ow + x;
    int sx = min(max(__float2int_rn(mapx[mi]), -32768), 32767);
    int sy = min(max(__float2int_rn(mapy[mi]), -32768), 32767);
    const bool inside = (unsigned)sx < (unsigned)w && (unsigned)sy < (unsigned)h;
    const double* p = depth + (size_t)z0 * w * h + (inside ? (size_t)sy * w + sx : 0);
    double* o = out + (size_t)z0 * ow * oh + mi;
    const double fx = (double)sx, fy = (double)sy;
#pragma unroll 4
    for (int z = 0; z < nz; z++, p += (size_t)w * h, o += (size_t)ow * oh) {
        double r = 0.;
        if (inside) {
            const double zz = *p;
            // row 2 of M @ [x*z, y*z, z] as NumPy's matmul (a BLAS dgemm with FMA kernels) rounds it:
            // fma(M22, z, fma(M21, y*z, M20*(x*z))) -- bit-identical with the reference's own run
            // (tests/golden/reference_plumbing.npz); individually rounded products and sums are 1 ulp off in 1 of 4 pixels
            r = __fma_rn(m2, zz, __fma_rn(m1, __dmul_rn(fy, zz), __dmul_rn(m0, __dmul_rn(fx, zz))));
        }
        *o = r;
    }
}

}  // namespace camd

using namespace camd;

extern "C" {

int camd_disp_to_depth(const int16_t* disp16, const uint8_t* valid_mask, int w, int h,
                       int sgbm_min_disparity, int add_min_disparity, int translate, double baseline_fx,
                       double max_depth, float* disparity, double* depth, int batch, void* stream)
{
    if (!disp16 || !valid_mask || !disparity || !depth || w <= 0 || h <= 0 || batch <= 0) {
        set_error("camd_disp_to_depth: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    int n = w * h;
    hipLaunchKernelGGL(k_disp_to_depth, dim3(div_up(div_up(n, 2), 256), batch), dim3(256), 0, (hipStream_t)stream, disp16,
                       valid_mask, n, (float)(sgbm_min_disparity * 16), (float)add_min_disparity, translate,
                       (float)w, baseline_fx, max_depth, disparity, depth);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_unrectify_depth(const double* depth, int w, int h, const double M[3], const float* mapx,
                         const float* mapy, double* out, int ow, int oh, int batch, void* stream)
{
    if (!depth || !M || !mapx || !mapy || !out || w <= 0 || h <= 0 || ow <= 0 || oh <= 0 || batch <= 0) {
        set_error("camd_unrectify_depth: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    int zb = batch < 16 ? batch : 16;  // all images of the batch (up to 16) per workgroup while the grid fills the chip
    while (zb > 1 && (long long)div_up(ow, 256) * oh * div_up(batch, zb) < 4096) zb = (zb + 1) / 2;
    hipLaunchKernelGGL(k_unrectify, dim3(div_up(ow, 256), oh, div_up(batch, zb)), dim3(256), 0, (hipStream_t)stream,
                       depth, w, h, M[0], M[1], M[2], mapx, mapy, out, ow, oh, batch, zb);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

}  // extern "C"

// resize.hip -- cv2.resize(..., INTER_LINEAR) for gfx950: the down-/up-scale boxx.resize performs around
// the matcher when max(h, w) > cfg["max_size"].
//
// Replaces (file:line in /root/reference/calibrating/):
//   stereo_matching.py:62   boxx.resize(img1|img2, resize_ratio)          u8 RGB, fixed point (11-bit)
//   stereo_matching.py:66   boxx.resize(sdisparity / 16.0, (h, w))        float32
// Arithmetic follows OpenCV's generic two-pass linear resize (resize.cpp): source position
// (d + 0.5) * scale - 0.5 evaluated in double and rounded to float, x index clamped with its weight
// zeroed, rows clamped; 8-bit: weights short(rint(w * 2048)), horizontal sums in int, vertical
// ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2; float: products and sums individually
// rounded.  Exact 2x decimation is the 2x2 box average (OpenCV switches to INTER_AREA there).
// One thread per destination pixel; the four taps come straight from HBM/L2 (the kernel is tiny).
#include "common.hpp"

namespace camd {

struct Axis {
    int s;     // source index (x: clamped)
    float f;   // fractional weight
    bool edge; // x only: s + 1 is outside -> single tap with full weight
};

__device__ __forceinline__ Axis axis_x(int d, double scale, int ssize)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    if (s < 0) { f = 0.f; s = 0; }
    bool edge = s + 1 >= ssize;
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    return Axis{s, f, edge};
}
__device__ __forceinline__ Axis axis_y(int d, double scale)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    return Axis{s, f, false};
}

// A block walks RESIZE_ROWS destination rows: what depends on the column only (cell, weights) is worked out once, and
// the grid has an eighth of the workgroups (one tiny workgroup per row was bound by the rate at which they start).
constexpr int RESIZE_ROWS = 8;

template <int CN>
__global__ __launch_bounds__(256) void k_resize_u8(const uint8_t* __restrict__ src, int sw, int sh,
                                                   uint8_t* __restrict__ dst, int dw, int dh, double scx,
                                                   double scy, int area2)
{
    const int x = blockIdx.x * 256 + threadIdx.x, ya = blockIdx.y * RESIZE_ROWS, yb = min(ya + RESIZE_ROWS, dh);
    if (x >= dw) return;
    const uint8_t* s = src + (size_t)blockIdx.z * sw * sh * CN;
    if (area2) {
        for (int y = ya; y < yb; y++) {
            uint8_t* o = dst + ((size_t)blockIdx.z * dh * dw + (size_t)y * dw + x) * CN;
            const uint8_t* p = s + ((size_t)(2 * y) * sw + 2 * x) * CN;
#pragma unroll
            for (int c = 0; c < CN; c++)
                o[c] = (uint8_t)((p[c] + p[CN + c] + p[(size_t)sw * CN + c] + p[(size_t)sw * CN + CN + c] + 2) >> 2);
        }
        return;
    }
    const Axis ax = axis_x(x, scx, sw);
    const int a0 = (short)__float2int_rn(__fmul_rn(__fsub_rn(1.f, ax.f), 2048.f));
    const int a1 = (short)__float2int_rn(__fmul_rn(ax.f, 2048.f));
    for (int y = ya; y < yb; y++) {
        uint8_t* o = dst + ((size_t)blockIdx.z * dh * dw + (size_t)y * dw + x) * CN;
        const Axis ay = axis_y(y, scy);
        const int sy0 = min(max(ay.s, 0), sh - 1), sy1 = min(max(ay.s + 1, 0), sh - 1);
        const int b0 = (short)__float2int_rn(__fmul_rn(__fsub_rn(1.f, ay.f), 2048.f));
        const int b1 = (short)__float2int_rn(__fmul_rn(ay.f, 2048.f));
        const uint8_t* r0 = s + ((size_t)sy0 * sw + ax.s) * CN;
        const uint8_t* r1 = s + ((size_t)sy1 * sw + ax.s) * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            int h0, h1;
            if (!ax.edge) { h0 = r0[c] * a0 + r0[CN + c] * a1; h1 = r1[c] * a0 + r1[CN + c] * a1; }
            else { h0 = r0[c] * 2048; h1 = r1[c] * 2048; }
            o[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

__global__ __launch_bounds__(256) void k_resize_f32(const float* __restrict__ src, int sw, int sh,
                                                    float* __restrict__ dst, int dw, int dh, double scx,
                                                    double scy, int area2)
{
    const int x = blockIdx.x * 256 + threadIdx.x, ya = blockIdx.y * RESIZE_ROWS, yb = min(ya + RESIZE_ROWS, dh);
    if (x >= dw) return;
    const float* s = src + (size_t)blockIdx.z * sw * sh;
    if (area2) {
        for (int y = ya; y < yb; y++) {
            const float* p = s + (size_t)(2 * y) * sw + 2 * x;
            dst[(size_t)blockIdx.z * dh * dw + (size_t)y * dw + x] =
                __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(p[0], p[1]), p[sw]), p[sw + 1]), 0.25f);
        }
        return;
    }
    const Axis ax = axis_x(x, scx, sw);
    const float a0 = __fsub_rn(1.f, ax.f), a1 = ax.f;
    for (int y = ya; y < yb; y++) {
        const Axis ay = axis_y(y, scy);
        const int sy0 = min(max(ay.s, 0), sh - 1), sy1 = min(max(ay.s + 1, 0), sh - 1);
        const float b0 = __fsub_rn(1.f, ay.f), b1 = ay.f;
        const float* r0 = s + (size_t)sy0 * sw + ax.s;
        const float* r1 = s + (size_t)sy1 * sw + ax.s;
        float h0, h1;
        if (!ax.edge) {
            h0 = __fadd_rn(__fmul_rn(r0[0], a0), __fmul_rn(r0[1], a1));
            h1 = __fadd_rn(__fmul_rn(r1[0], a0), __fmul_rn(r1[1], a1));
        } else { h0 = r0[0]; h1 = r1[0]; }
        dst[(size_t)blockIdx.z * dh * dw + (size_t)y * dw + x] = __fadd_rn(__fmul_rn(h0, b0), __fmul_rn(h1, b1));
    }
}

// The matcher's downsizing branch in one pass (stereo_matching.py:63-69 with max_size < image, stereo_camera.py:510-513,
// :408-413): the int16 disparity of the DOWNSIZED pair -> float32, clip at 0, below minDisparity*16 -> 0, /16 (all
// on the fly, per source tap), cv2.resize(INTER_LINEAR) up to the rectified size, * w / sw, += min_disparity, * mask,
// depth = baseline*fx / disparity with the two clamps.  Replaces k_resize_f32 and a dozen elementwise launches; every
// float operation is the one NumPy / cv2 performs, in their order.
__global__ __launch_bounds__(256) void k_disp16_up_to_depth(const int16_t* __restrict__ src, int sw, int sh,
                                                            const uint8_t* __restrict__ mask, int dw, int dh,
                                                            double scx, double scy, float thresh, float addv,
                                                            int translate, float wf, float swf, double bf,
                                                            double max_depth, float* __restrict__ disparity,
                                                            double* __restrict__ depth)
{
    const int x = blockIdx.x * 256 + threadIdx.x, ya = blockIdx.y * RESIZE_ROWS, yb = min(ya + RESIZE_ROWS, dh);
    if (x >= dw) return;
    const int16_t* s = src + (size_t)blockIdx.z * sw * sh;
    auto tap = [&](const int16_t* p) {
        float v = (float)*p;
        v = v < 0.f ? 0.f : v;
        v = v < thresh ? 0.f : v;
        return v / 16.0f;
    };
    const Axis ax = axis_x(x, scx, sw);
    const float a0 = __fsub_rn(1.f, ax.f), a1 = ax.f;
    for (int y = ya; y < yb; y++) {
        const Axis ay = axis_y(y, scy);
        const int sy0 = min(max(ay.s, 0), sh - 1), sy1 = min(max(ay.s + 1, 0), sh - 1);
        const float b0 = __fsub_rn(1.f, ay.f), b1 = ay.f;
        const int16_t* r0 = s + (size_t)sy0 * sw + ax.s;
        const int16_t* r1 = s + (size_t)sy1 * sw + ax.s;
        float h0, h1;
        if (!ax.edge) {
            h0 = __fadd_rn(__fmul_rn(tap(r0), a0), __fmul_rn(tap(r0 + 1), a1));
            h1 = __fadd_rn(__fmul_rn(tap(r1), a0), __fmul_rn(tap(r1 + 1), a1));
        } else { h0 = tap(r0); h1 = tap(r1); }
        float d = __fadd_rn(__fmul_rn(h0, b0), __fmul_rn(h1, b1));
        d = __fdiv_rn(__fmul_rn(d, wf), swf);  // * w / sw, each op rounded like NumPy
        if (translate) d = __fadd_rn(d, addv);
        const size_t i = (size_t)y * dw + x, o = (size_t)blockIdx.z * dw * dh + i;
        d = mask[i] ? d : __fmul_rn(0.f, d);
        disparity[o] = d;
        double z = __ddiv_rn(bf, (double)d);
        z = z > max_depth ? 0. : z;
        depth[o] = z < 0. ? 0. : z;
    }
}

}  // namespace camd

using namespace camd;

extern "C" {

int camd_resize_linear_u8(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh, int batch,
                          void* stream)
{
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0 || (cn != 1 && cn != 3)) {
        set_error("camd_resize_linear_u8: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (sw == dw && sh == dh) {
        CAMD_HIP(hipMemcpyAsync(dst, src, (size_t)batch * sw * sh * cn, hipMemcpyDeviceToDevice, st));
        return CAMD_OK;
    }
    const int area2 = sw == dw * 2 && sh == dh * 2;
    const double scx = (double)sw / dw, scy = (double)sh / dh;
    dim3 grid(div_up(dw, 256), div_up(dh, RESIZE_ROWS), batch), block(256);
    if (cn == 1) hipLaunchKernelGGL((k_resize_u8<1>), grid, block, 0, st, src, sw, sh, dst, dw, dh, scx, scy, area2);
    else hipLaunchKernelGGL((k_resize_u8<3>), grid, block, 0, st, src, sw, sh, dst, dw, dh, scx, scy, area2);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh, int batch, void* stream)
{
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0) {
        set_error("camd_resize_linear_f32: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (sw == dw && sh == dh) {
        CAMD_HIP(hipMemcpyAsync(dst, src, (size_t)batch * sw * sh * 4, hipMemcpyDeviceToDevice, st));
        return CAMD_OK;
    }
    const int area2 = sw == dw * 2 && sh == dh * 2;
    const double scx = (double)sw / dw, scy = (double)sh / dh;
    hipLaunchKernelGGL(k_resize_f32, dim3(div_up(dw, 256), div_up(dh, RESIZE_ROWS), batch), dim3(256), 0, st, src, sw, sh, dst, dw, dh,
                       scx, scy, area2);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_disp16_resized_to_depth(const int16_t* disp16, int sw, int sh, const uint8_t* valid_mask, int w, int h,
                                 int sgbm_min_disparity, int add_min_disparity, int translate, double baseline_fx,
                                 double max_depth, float* disparity, double* depth, int batch, void* stream)
{
    if (!disp16 || !valid_mask || !disparity || !depth || sw <= 0 || sh <= 0 || w <= 0 || h <= 0 || batch <= 0) {
        set_error("camd_disp16_resized_to_depth: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    if (sw == w * 2 && sh == h * 2) {  // (cv2.resize takes its 2x2 area path there; the matcher never upsizes by 1/2)
        set_error("camd_disp16_resized_to_depth: exact 2:1 reduction is not a case of the matcher's resize back");
        return CAMD_ERR_UNSUPPORTED;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    hipLaunchKernelGGL(k_disp16_up_to_depth, dim3(div_up(w, 256), div_up(h, RESIZE_ROWS), batch), dim3(256), 0, (hipStream_t)stream, disp16, sw,
                       sh, valid_mask, w, h, (double)sw / w, (double)sh / h, (float)(sgbm_min_disparity * 16),
                       (float)add_min_disparity, translate, (float)w, (float)sw, baseline_fx, max_depth, disparity,
                       depth);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

}  // extern "C"

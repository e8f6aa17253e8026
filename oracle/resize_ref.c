/*
 * resize_ref.c -- scalar CPU restatement of cv2.resize(..., INTER_LINEAR) as boxx.resize calls it.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: restates the generic C++ path of
 * opencv/modules/imgproc/src/resize.cpp (resizeGeneric_ + HResizeLinear + VResizeLinear, and the
 * exact-2x shortcut to the INTER_AREA box average), from the published OpenCV 4.x source; cv2 wheels
 * may route 8-bit linear resizes through IPP, whose rounding can differ by 1 LSB.
 * Reference call sites: /root/reference/calibrating/stereo_matching.py:62 (downsize of the RGB pair to
 * max_size) and :66 (upsize of the float32 disparity), both through boxx.resize -> cv2.resize.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

enum { COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS };

static inline int clipi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* per-axis tables: source index (clamped like OpenCV does for the x axis), fractional weight */
static void axis_table(int dsize, int ssize, int clamp_x, int* ofs, float* frac, int* dmax)
{
    double scale = (double)ssize / dsize;
    *dmax = dsize;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (clamp_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s + 1 >= ssize) {
                if (d < *dmax) *dmax = d;
                if (s >= ssize - 1) { f = 0; s = ssize - 1; }
            }
        }
        ofs[d] = s;
        frac[d] = f;
    }
}

int oracle_resize_linear_u8(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh)
{
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    if (sw == dw && sh == dh) { /* same size: cv2 copies */
        for (size_t i = 0; i < (size_t)sw * sh * cn; i++) dst[i] = src[i];
        return 0;
    }
    if (sw == dw * 2 && sh == dh * 2) { /* exact 2x decimation: INTER_LINEAR == fast INTER_AREA */
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const uint8_t* p = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
                    dst[((size_t)y * dw + x) * cn + c] =
                        (uint8_t)((p[0] + p[cn] + p[(size_t)sw * cn] + p[(size_t)sw * cn + cn] + 2) >> 2);
                }
        return 0;
    }
    int *xofs = malloc(sizeof(int) * dw), *yofs = malloc(sizeof(int) * dh), xmax, ymax;
    float *fx = malloc(sizeof(float) * dw), *fy = malloc(sizeof(float) * dh);
    axis_table(dw, sw, 1, xofs, fx, &xmax);
    axis_table(dh, sh, 0, yofs, fy, &ymax);
    for (int y = 0; y < dh; y++) {
        int sy0 = clipi(yofs[y], 0, sh - 1), sy1 = clipi(yofs[y] + 1, 0, sh - 1);
        int b0 = (short)lrintf((1.f - fy[y]) * COEF_SCALE), b1 = (short)lrintf(fy[y] * COEF_SCALE);
        for (int x = 0; x < dw; x++) {
            int a0 = (short)lrintf((1.f - fx[x]) * COEF_SCALE), a1 = (short)lrintf(fx[x] * COEF_SCALE);
            for (int c = 0; c < cn; c++) {
                const uint8_t* r0 = src + ((size_t)sy0 * sw + xofs[x]) * cn + c;
                const uint8_t* r1 = src + ((size_t)sy1 * sw + xofs[x]) * cn + c;
                int h0, h1;
                if (x < xmax) { h0 = r0[0] * a0 + r0[cn] * a1; h1 = r1[0] * a0 + r1[cn] * a1; }
                else { h0 = r0[0] * COEF_SCALE; h1 = r1[0] * COEF_SCALE; }
                dst[((size_t)y * dw + x) * cn + c] =
                    (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xofs); free(yofs); free(fx); free(fy);
    return 0;
}

int oracle_resize_linear_f32(const float* src, int sw, int sh, float* dst, int dw, int dh)
{
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    if (sw == dw && sh == dh) {
        for (size_t i = 0; i < (size_t)sw * sh; i++) dst[i] = src[i];
        return 0;
    }
    if (sw == dw * 2 && sh == dh * 2) { /* fast INTER_AREA on floats: (a+b+c+d) * 0.25 */
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                const float* p = src + (size_t)(2 * y) * sw + 2 * x;
                dst[(size_t)y * dw + x] = (p[0] + p[1] + p[sw] + p[sw + 1]) * 0.25f;
            }
        return 0;
    }
    int *xofs = malloc(sizeof(int) * dw), *yofs = malloc(sizeof(int) * dh), xmax, ymax;
    float *fx = malloc(sizeof(float) * dw), *fy = malloc(sizeof(float) * dh);
    axis_table(dw, sw, 1, xofs, fx, &xmax);
    axis_table(dh, sh, 0, yofs, fy, &ymax);
    for (int y = 0; y < dh; y++) {
        int sy0 = clipi(yofs[y], 0, sh - 1), sy1 = clipi(yofs[y] + 1, 0, sh - 1);
        float b0 = 1.f - fy[y], b1 = fy[y];
        for (int x = 0; x < dw; x++) {
            float a0 = 1.f - fx[x], a1 = fx[x];
            const float* r0 = src + (size_t)sy0 * sw + xofs[x];
            const float* r1 = src + (size_t)sy1 * sw + xofs[x];
            volatile float h0, h1; /* volatile: keep every product/sum individually rounded (no FMA) */
            if (x < xmax) {
                volatile float p00 = r0[0] * a0, p01 = r0[1] * a1, p10 = r1[0] * a0, p11 = r1[1] * a1;
                h0 = p00 + p01; h1 = p10 + p11;
            } else { h0 = r0[0]; h1 = r1[0]; }
            volatile float q0 = h0 * b0, q1 = h1 * b1;
            dst[(size_t)y * dw + x] = q0 + q1;
        }
    }
    free(xofs); free(yofs); free(fx); free(fy);
    return 0;
}

/*
 * sgbm_ref.c -- scalar CPU restatement of cv::StereoSGBM::compute (modes MODE_SGBM, MODE_HH, MODE_HH4).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: OpenCV is a third-party dependency
 * of the reference (`opencv-contrib-python>=4.7.0.72`, /root/reference/requirements.txt:2) that is
 * not vendored and not importable where this was written.  The code below restates the published
 * algorithm of opencv/modules/calib3d/src/stereosgbm.cpp (4.x: calcPixelCostBT,
 * computeDisparitySGBM, StereoSGBMImpl::compute, filterSpecklesImpl) and of
 * opencv/modules/imgproc/src/median_blur.simd.hpp (medianBlur_SortNet, ksize 3) in their scalar
 * (non-SIMD) form, deliberately keeping OpenCV's row-incremental structure (hsum ring buffer,
 * two-row Lr buffers, in-loop fifth path) so that the GPU's parallel reformulation is checked
 * against the sequential original.  Reference call site:
 *   /root/reference/calibrating/stereo_matching.py:48-58 (StereoSGBM_create, default mode)
 *   /root/reference/calibrating/stereo_matching.py:63    (.compute(simg1, simg2))
 */
#include "oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

typedef int16_t CostType;
typedef int16_t DispType;
typedef uint8_t PixType;

enum { DISP_SHIFT = 4, DISP_SCALE = 1 << DISP_SHIFT };
#define MAX_COST ((CostType)SHRT_MAX)

static oracle_switches g_sw = {4, 1, 1, 4, 8};
void oracle_set_switches(const oracle_switches* s) { g_sw = *s; }
void oracle_get_switches(oracle_switches* s) { *s = g_sw; }

/* U7: one int16 add / sub / mul of the cost-volume recurrences -- saturating (v_int16 operators of OpenCV's
 * SIMD path) or wrapping (the scalar path's (CostType) cast) */
static inline short cost_wrap(int v) { return (short)v; }
static inline short cost_sat(int v) { return (short)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }
static inline short c_add(int a, int b) { return g_sw.cost_saturate ? cost_sat(a + b) : cost_wrap(a + b); }
static inline short c_sub(int a, int b) { return g_sw.cost_saturate ? cost_sat(a - b) : cost_wrap(a - b); }
static inline short c_mul(int a, int b) { return g_sw.cost_saturate ? cost_sat(a * b) : cost_wrap(a * b); }
int oracle_sw_lanczos_group(void) { return g_sw.lanczos_fix_group_lo; }

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline CostType sat16(int v) { return (CostType)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v); }

/* ---- stereosgbm.cpp: calcPixelCostBT --------------------------------------------------------
 * Birchfield-Tomasi cost of row y for every (x, d), summed over the 2*cn planes
 * (x-Sobel clipped to [0, 2*ftzero] and raw intensity with cost >> 2).
 * cost layout: cost[(x - minX1)*D + (d - minD)].
 * buffer: width2*2 + width*cn*4 bytes of scratch.                                              */
static void calc_pixel_cost_bt(const PixType* img1, const PixType* img2, size_t step, int width,
                               int height, int cn, int y, int minD, int maxD, CostType* cost,
                               PixType* buffer, const PixType* tab)
{
    int x, c;
    int minX1 = imax(maxD, 0), maxX1 = width + imin(minD, 0);
    int D = maxD - minD, width1 = maxX1 - minX1;
    int minX2 = imax(minX1 - maxD, 0), maxX2 = imin(maxX1 - minD, width);
    int width2 = maxX2 - minX2;
    const PixType* row1 = img1 + (size_t)y * step;
    const PixType* row2 = img2 + (size_t)y * step;
    PixType *prow1 = buffer + width2 * 2, *prow2 = prow1 + width * cn * 2;

    for (c = 0; c < cn * 2; c++) {
        /* columns 0 and width-1 of EVERY plane hold tab[0] (= ftzero) */
        if (c < cn || g_sw.bt_border_raw_tab0) {
            prow1[width * c] = prow1[width * c + width - 1] = prow2[width * c] =
                prow2[width * c + width - 1] = tab[0];
        } else {
            prow1[width * c] = row1[(size_t)0 * cn + (c - cn)];
            prow1[width * c + width - 1] = row1[(size_t)(width - 1) * cn + (c - cn)];
            prow2[width * c + width - 1] = row2[(size_t)0 * cn + (c - cn)];
            prow2[width * c] = row2[(size_t)(width - 1) * cn + (c - cn)];
        }
    }

    ptrdiff_t n1 = y > 0 ? -(ptrdiff_t)step : 0, s1 = y < height - 1 ? (ptrdiff_t)step : 0;
    ptrdiff_t n2 = n1, s2 = s1;

    if (cn == 1) {
        for (x = 1; x < width - 1; x++) {
            prow1[x] = tab[(row1[x + 1] - row1[x - 1]) * 2 + row1[x + n1 + 1] - row1[x + n1 - 1] +
                           row1[x + s1 + 1] - row1[x + s1 - 1]];
            prow2[width - 1 - x] = tab[(row2[x + 1] - row2[x - 1]) * 2 + row2[x + n2 + 1] -
                                       row2[x + n2 - 1] + row2[x + s2 + 1] - row2[x + s2 - 1]];
            prow1[x + width] = row1[x];
            prow2[width - 1 - x + width] = row2[x];
        }
    } else {
        for (x = 1; x < width - 1; x++) {
            for (c = 0; c < cn; c++) {
                const PixType* r1 = row1 + x * cn + c;
                const PixType* r2 = row2 + x * cn + c;
                prow1[x + width * c] = tab[(r1[cn] - r1[-cn]) * 2 + r1[n1 + cn] - r1[n1 - cn] +
                                           r1[s1 + cn] - r1[s1 - cn]];
                prow2[width - 1 - x + width * c] = tab[(r2[cn] - r2[-cn]) * 2 + r2[n2 + cn] -
                                                       r2[n2 - cn] + r2[s2 + cn] - r2[s2 - cn]];
                prow1[x + width * (cn + c)] = r1[0];
                prow2[width - 1 - x + width * (cn + c)] = r2[0];
            }
        }
    }

    memset(cost, 0, (size_t)width1 * D * sizeof(cost[0]));

    buffer -= width - maxX2;
    cost -= minX1 * D + minD; /* simplify the cost indices inside the loop */

    for (c = 0; c < cn * 2; c++, prow1 += width, prow2 += width) {
        int diff_scale = c < cn ? 0 : 2;

        /* v0 = min(row2[x-1/2], row2[x], row2[x+1/2]), v1 = max(...) on the x-reversed right row */
        for (x = width - maxX2; x < width - minX2; x++) {
            int v = prow2[x];
            int vl = x > 0 ? (v + prow2[x - 1]) / 2 : v;
            int vr = x < width - 1 ? (v + prow2[x + 1]) / 2 : v;
            int v0 = imin(vl, vr);
            v0 = imin(v0, v);
            int v1 = imax(vl, vr);
            v1 = imax(v1, v);
            buffer[x] = (PixType)v0;
            buffer[x + width2] = (PixType)v1;
        }

        for (x = minX1; x < maxX1; x++) {
            int u = prow1[x];
            int ul = x > 0 ? (u + prow1[x - 1]) / 2 : u;
            int ur = x < width - 1 ? (u + prow1[x + 1]) / 2 : u;
            int u0 = imin(ul, ur);
            u0 = imin(u0, u);
            int u1 = imax(ul, ur);
            u1 = imax(u1, u);

            for (int d = minD; d < maxD; d++) {
                int v = prow2[width - x - 1 + d];
                int v0 = buffer[width - x - 1 + d];
                int v1 = buffer[width - x - 1 + d + width2];
                int c0 = imax(0, u - v1);
                c0 = imax(c0, v0 - u);
                int c1 = imax(0, v - u1);
                c1 = imax(c1, u0 - v);
                cost[x * D + d] = (CostType)(cost[x * D + d] + (imin(c0, c1) >> diff_scale));
            }
        }
    }
}

/* what to capture besides the disparity (stage-wise parity) */
typedef struct capture {
    CostType* C; /* [height][width1][D] or NULL */
    CostType* S; /* [height][width1][D] or NULL */
} capture;

/* ---- stereosgbm.cpp: computeDisparitySGBM ----------------------------------------------------
 * disp1: int16 [height][width].  Returns 0 or -1 (allocation failure).                          */
static int compute_disparity_sgbm(const PixType* img1, const PixType* img2, int width, int height,
                                  int cn, size_t step, DispType* disp1,
                                  const oracle_sgbm_params* params, const capture* cap)
{
    int minD = params->minDisparity, maxD = minD + params->numDisparities;
    int uniquenessRatio = params->uniquenessRatio >= 0 ? params->uniquenessRatio : 10;
    int disp12MaxDiff = params->disp12MaxDiff > 0 ? params->disp12MaxDiff : 1;
    int P1 = params->P1 > 0 ? params->P1 : 2;
    int P2 = imax(params->P2 > 0 ? params->P2 : 5, P1 + 1);
    int k;
    int minX1 = imax(maxD, 0), maxX1 = width + imin(minD, 0);
    const int D = params->numDisparities;
    int width1 = maxX1 - minX1;
    int INVALID_DISP = minD - 1, INVALID_DISP_SCALED = INVALID_DISP * DISP_SCALE;
    int SADWindowSize = params->blockSize > 0 ? params->blockSize : 5;
    int SW2 = SADWindowSize / 2, SH2 = SADWindowSize / 2;
    int fullDP = params->mode == 1;
    int npasses = fullDP ? 2 : 1;
    const int TAB_OFS = 256 * 4, TAB_SIZE = 256 + TAB_OFS * 2;
    PixType clipTab[256 + 256 * 4 * 2];
    int ftzero = imax(params->preFilterCap, 15) | 1;

    for (k = 0; k < TAB_SIZE; k++)
        clipTab[k] = (PixType)(imin(imax(k - TAB_OFS, -ftzero), ftzero) + ftzero);

    if (minX1 >= maxX1) {
        for (size_t i = 0; i < (size_t)width * height; i++) disp1[i] = (DispType)INVALID_DISP_SCALED;
        return 0;
    }
    /* The first box sum below reads pixDiff columns 0..SW2 unclamped (as cv2 does): with width1 <= SW2 cv2
     * reads memory it never wrote, so no reference answer exists -- refuse instead of over-reading. */
    if (width1 <= SW2) return -2;

    /* buffers (BufferSGBM) */
    const int NR = 4;            /* directions per pass */
    const int D2 = D + 2;        /* guard slots d = -1 and d = D */
    const int LrBorder = 1;      /* one border pixel left and right */
    size_t costWidth = (size_t)width1 * D;
    size_t costHeight = fullDP ? (size_t)height : 1;
    int hsumRows = SH2 * 2 + 2;
    size_t LrSize = (size_t)(width1 + LrBorder * 2) * NR * D2;
    size_t minLrSize = (size_t)(width1 + LrBorder * 2) * NR;

    CostType* Cbuf = (CostType*)malloc(costWidth * costHeight * sizeof(CostType));
    CostType* Sbuf = (CostType*)malloc(costWidth * costHeight * sizeof(CostType));
    CostType* hsumBuf = (CostType*)malloc(costWidth * hsumRows * sizeof(CostType));
    CostType* pixDiff = (CostType*)malloc(costWidth * sizeof(CostType));
    CostType* LrBuf[2] = {(CostType*)malloc(LrSize * sizeof(CostType)),
                          (CostType*)malloc(LrSize * sizeof(CostType))};
    CostType* minLrBuf[2] = {(CostType*)malloc(minLrSize * sizeof(CostType)),
                             (CostType*)malloc(minLrSize * sizeof(CostType))};
    CostType* disp2cost = (CostType*)malloc((size_t)width * sizeof(CostType));
    DispType* disp2ptr = (DispType*)malloc((size_t)width * sizeof(DispType));
    PixType* tempBuf = (PixType*)malloc((size_t)width * (4 * cn + 2) + 64);
    if (!Cbuf || !Sbuf || !hsumBuf || !pixDiff || !LrBuf[0] || !LrBuf[1] || !minLrBuf[0] ||
        !minLrBuf[1] || !disp2cost || !disp2ptr || !tempBuf) {
        free(Cbuf); free(Sbuf); free(hsumBuf); free(pixDiff); free(LrBuf[0]); free(LrBuf[1]);
        free(minLrBuf[0]); free(minLrBuf[1]); free(disp2cost); free(disp2ptr); free(tempBuf);
        return -1;
    }

#define GET_C(row) (Cbuf + (fullDP ? (size_t)(row) * costWidth : 0))
#define GET_S(row) (Sbuf + (fullDP ? (size_t)(row) * costWidth : 0))
#define GET_HSUM(row) (hsumBuf + (size_t)((row) % hsumRows) * costWidth)
/* Lr(id, x, dir): x in [-1, width1]; points at d = 0, slots [-1] and [D] are the guards */
#define GET_LR(id, x, dir) (LrBuf[id] + ((size_t)((x) + LrBorder) * NR + (dir)) * D2 + 1)
#define GET_MINLR(id, x, dir) (minLrBuf[id] + (size_t)((x) + LrBorder) * NR + (dir))

    /* mem.initCBuf(P2): add P2 to every C(x,y); it saves a few operations in the inner loops */
    for (size_t i = 0; i < costWidth * costHeight; i++) Cbuf[i] = (CostType)P2;

    for (int pass = 1; pass <= npasses; pass++) {
        int x1, y1, x2, y2, dx, dy;

        if (pass == 1) {
            y1 = 0; y2 = height; dy = 1;
            x1 = 0; x2 = width1; dx = 1;
        } else {
            y1 = height - 1; y2 = -1; dy = -1;
            x1 = width1 - 1; x2 = -1; dx = -1;
        }

        int lrID = 0;
        /* mem.clearLr(): both row buffers incl. borders are zero at the start of each pass */
        memset(LrBuf[0], 0, LrSize * sizeof(CostType));
        memset(LrBuf[1], 0, LrSize * sizeof(CostType));
        memset(minLrBuf[0], 0, minLrSize * sizeof(CostType));
        memset(minLrBuf[1], 0, minLrSize * sizeof(CostType));

        for (int y = y1; y != y2; y += dy) {
            int x, d;
            DispType* disp1ptr = disp1 + (size_t)y * width;
            CostType* const C = GET_C(y);
            CostType* const S = GET_S(y);

            if (pass == 1) /* compute C on the first pass, and reuse it on the second pass, if any */
            {
                int dy1 = y == 0 ? 0 : y + SH2, dy2 = y == 0 ? SH2 : dy1;

                for (k = dy1; k <= dy2; k++) {
                    CostType* hsumAdd = GET_HSUM(imin(k, height - 1));

                    if (k < height) {
                        calc_pixel_cost_bt(img1, img2, step, width, height, cn, k, minD, maxD,
                                           pixDiff, tempBuf, clipTab + TAB_OFS);

                        /* Operation order as in OpenCV's CV_SIMD branches (it only matters when cost_saturate
                         * is on and a value crosses 32767): hsumAdd[0] = pix[0]*(SW2+1) + pix[1] + ...;
                         * hv = hsumAdd[x-1] - pixSub + pixAdd; column 0: C = Cprev + hsumAdd - hsumSub;
                         * columns >= 1: C = Cprev - hsumSub + hv; first row: C = C + hv*scale. */
                        for (d = 0; d < D; d++) {
                            CostType h = c_mul(pixDiff[d], SW2 + 1);
                            for (x = D; x <= SW2 * D; x += D) h = c_add(h, pixDiff[x + d]);
                            hsumAdd[d] = h;
                        }

                        if (y > 0) {
                            const CostType* hsumSub = GET_HSUM(imax(y - SH2 - 1, 0));
                            const CostType* Cprev = GET_C(y - 1);

                            for (d = 0; d < D; d++)
                                C[d] = c_sub(c_add(Cprev[d], hsumAdd[d]), hsumSub[d]);

                            for (x = D; x < width1 * D; x += D) {
                                const CostType* pixAdd = pixDiff + imin(x + SW2 * D, (width1 - 1) * D);
                                const CostType* pixSub = pixDiff + imax(x - (SW2 + 1) * D, 0);
                                for (d = 0; d < D; d++) {
                                    CostType hv = c_add(c_sub(hsumAdd[x - D + d], pixSub[d]), pixAdd[d]);
                                    hsumAdd[x + d] = hv;
                                    C[x + d] = c_add(c_sub(Cprev[x + d], hsumSub[x + d]), hv);
                                }
                            }
                        } else {
                            int scale = k == 0 ? SH2 + 1 : 1;
                            for (d = 0; d < D; d++)
                                C[d] = c_add(C[d], c_mul(hsumAdd[d], scale));
                            for (x = D; x < width1 * D; x += D) {
                                const CostType* pixAdd = pixDiff + imin(x + SW2 * D, (width1 - 1) * D);
                                const CostType* pixSub = pixDiff + imax(x - (SW2 + 1) * D, 0);
                                for (d = 0; d < D; d++) {
                                    CostType hv = c_add(c_sub(hsumAdd[x - D + d], pixSub[d]), pixAdd[d]);
                                    hsumAdd[x + d] = hv;
                                    C[x + d] = c_add(C[x + d], c_mul(hv, scale));
                                }
                            }
                        }
                    } else {
                        if (y > 0) {
                            const CostType* hsumSub = GET_HSUM(imax(y - SH2 - 1, 0));
                            const CostType* Cprev = GET_C(y - 1);
                            for (x = 0; x < width1 * D; x++)
                                C[x] = c_add(c_sub(Cprev[x], hsumSub[x]), hsumAdd[x]);
                        } else {
                            for (x = 0; x < width1 * D; x++)
                                C[x] = c_add(C[x], hsumAdd[x]);
                        }
                    }
                }

                /* also, clear the S buffer */
                memset(S, 0, costWidth * sizeof(CostType));
                if (cap && cap->C)
                    memcpy(cap->C + (size_t)y * costWidth, C, costWidth * sizeof(CostType));
            }

            /*
             [formula 13 in the paper]
             compute L_r(p, d) = C(p, d) +
                 min(L_r(p-r, d), L_r(p-r, d-1) + P1, L_r(p-r, d+1) + P1, min_k L_r(p-r, k) + P2)
                 - min_k L_r(p-r, k)
             where p = (x,y), r is one of the directions. All 4 directions of a pass at once:
                 0: r=(-dx, 0)   1: r=(-1, -dy)   2: r=(0, -dy)   3: r=(1, -dy)
            */
            for (x = x1; x != x2; x += dx) {
                int delta0 = P2 + *GET_MINLR(lrID, x - dx, 0);
                int delta1 = P2 + *GET_MINLR(1 - lrID, x - 1, 1);
                int delta2 = P2 + *GET_MINLR(1 - lrID, x, 2);
                int delta3 = P2 + *GET_MINLR(1 - lrID, x + 1, 3);

                CostType* Lr_p0 = GET_LR(lrID, x - dx, 0);
                CostType* Lr_p1 = GET_LR(1 - lrID, x - 1, 1);
                CostType* Lr_p2 = GET_LR(1 - lrID, x, 2);
                CostType* Lr_p3 = GET_LR(1 - lrID, x + 1, 3);

                Lr_p0[-1] = Lr_p0[D] = MAX_COST;
                Lr_p1[-1] = Lr_p1[D] = MAX_COST;
                Lr_p2[-1] = Lr_p2[D] = MAX_COST;
                Lr_p3[-1] = Lr_p3[D] = MAX_COST;

                CostType* Lr_p = GET_LR(lrID, x, 0);
                const CostType* Cp = C + (size_t)x * D;
                CostType* Sp = S + (size_t)x * D;
                int minL0 = MAX_COST, minL1 = MAX_COST, minL2 = MAX_COST, minL3 = MAX_COST;

                for (d = 0; d < D; d++) {
                    int Cpd = Cp[d], L0, L1, L2, L3;

                    L0 = Cpd + imin((int)Lr_p0[d], imin(Lr_p0[d - 1] + P1, imin(Lr_p0[d + 1] + P1, delta0))) - delta0;
                    L1 = Cpd + imin((int)Lr_p1[d], imin(Lr_p1[d - 1] + P1, imin(Lr_p1[d + 1] + P1, delta1))) - delta1;
                    L2 = Cpd + imin((int)Lr_p2[d], imin(Lr_p2[d - 1] + P1, imin(Lr_p2[d + 1] + P1, delta2))) - delta2;
                    L3 = Cpd + imin((int)Lr_p3[d], imin(Lr_p3[d - 1] + P1, imin(Lr_p3[d + 1] + P1, delta3))) - delta3;

                    Lr_p[d] = (CostType)L0;
                    minL0 = imin(minL0, L0);
                    Lr_p[d + D2] = (CostType)L1;
                    minL1 = imin(minL1, L1);
                    Lr_p[d + D2 * 2] = (CostType)L2;
                    minL2 = imin(minL2, L2);
                    Lr_p[d + D2 * 3] = (CostType)L3;
                    minL3 = imin(minL3, L3);

                    Sp[d] = sat16(Sp[d] + L0 + L1 + L2 + L3);
                }
                CostType* minL = GET_MINLR(lrID, x, 0);
                minL[0] = (CostType)minL0;
                minL[1] = (CostType)minL1;
                minL[2] = (CostType)minL2;
                minL[3] = (CostType)minL3;
            }

            if (pass == npasses) {
                for (x = 0; x < width; x++) {
                    disp1ptr[x] = disp2ptr[x] = (DispType)INVALID_DISP_SCALED;
                    disp2cost[x] = MAX_COST;
                }

                for (x = width1 - 1; x >= 0; x--) {
                    CostType* Sp = S + (size_t)x * D;
                    int minS = MAX_COST, bestDisp = -1;

                    if (npasses == 1) {
                        /* fifth path r=(+1,0): reuses the direction-0 slots of the current row */
                        CostType* Lr_p0 = GET_LR(lrID, x + 1, 0);
                        Lr_p0[-1] = Lr_p0[D] = MAX_COST;
                        CostType* Lr_p = GET_LR(lrID, x, 0);
                        const CostType* Cp = C + (size_t)x * D;
                        int delta0 = P2 + *GET_MINLR(lrID, x + 1, 0);
                        int minL0 = MAX_COST;

                        for (d = 0; d < D; d++) {
                            int L0 = Cp[d] + imin((int)Lr_p0[d], imin(Lr_p0[d - 1] + P1, imin(Lr_p0[d + 1] + P1, delta0))) - delta0;

                            Lr_p[d] = (CostType)L0;
                            minL0 = imin(minL0, L0);

                            int Sval = Sp[d] = sat16(Sp[d] + L0);
                            if (Sval < minS) {
                                minS = Sval;
                                bestDisp = d;
                            }
                        }
                        *GET_MINLR(lrID, x, 0) = (CostType)minL0;
                    } else {
                        for (d = 0; d < D; d++) {
                            int Sval = Sp[d];
                            if (Sval < minS) {
                                minS = Sval;
                                bestDisp = d;
                            }
                        }
                    }

                    for (d = 0; d < D; d++) {
                        if (Sp[d] * (100 - uniquenessRatio) < minS * 100 && abs(bestDisp - d) > 1)
                            break;
                    }
                    if (d < D)
                        continue;
                    d = bestDisp;
                    int _x2 = x + minX1 - d - minD;
                    if (disp2cost[_x2] > minS) {
                        disp2cost[_x2] = (CostType)minS;
                        disp2ptr[_x2] = (DispType)(d + minD);
                    }

                    if (0 < d && d < D - 1) {
                        /* subpixel: fit a parabola through (d-1,Sp[d-1]), (d,Sp[d]), (d+1,Sp[d+1]) */
                        int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                        d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
                    } else
                        d *= DISP_SCALE;
                    disp1ptr[x + minX1] = (DispType)(d + minD * DISP_SCALE);
                }

                for (x = minX1; x < maxX1; x++) {
                    /* round the disparity both towards -inf and +inf and check whether either of
                       the corresponding disparities in disp2 is consistent */
                    int d1 = disp1ptr[x];
                    if (d1 == INVALID_DISP_SCALED)
                        continue;
                    int _d = d1 >> DISP_SHIFT;
                    int d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
                    int _x = x - _d, x_ = x - d_;
                    if (0 <= _x && _x < width && disp2ptr[_x] >= minD && abs(disp2ptr[_x] - _d) > disp12MaxDiff &&
                        0 <= x_ && x_ < width && disp2ptr[x_] >= minD && abs(disp2ptr[x_] - d_) > disp12MaxDiff)
                        disp1ptr[x] = (DispType)INVALID_DISP_SCALED;
                }
                if (cap && cap->S)
                    memcpy(cap->S + (size_t)y * costWidth, S, costWidth * sizeof(CostType));
            }

            /* now shift the cyclic buffers */
            lrID = 1 - lrID;
        }
    }

    free(Cbuf); free(Sbuf); free(hsumBuf); free(pixDiff); free(LrBuf[0]); free(LrBuf[1]);
    free(minLrBuf[0]); free(minLrBuf[1]); free(disp2cost); free(disp2ptr); free(tempBuf);
    return 0;
#undef GET_C
#undef GET_S
#undef GET_HSUM
#undef GET_LR
#undef GET_MINLR
}

/* ---- stereosgbm.cpp: computeDisparitySGBM_HH4 (MODE_HH4 = 3) -----------------------------------
 * Four paths: a vertical stage (CalcVerticalSums: every column top-down then bottom-up, S = L_down, then
 * S += L_up) and a horizontal stage (CalcHorizontalSums: every row left-to-right, then right-to-left with
 * the winner-take-all / uniqueness / disp2 / sub-pixel tail in the same loop, then the left-right check).
 * C(x,y,d) is the same box-filtered BT cost (+P2) as in computeDisparitySGBM, taken from it by capture. */
static void lr_update(const CostType* Cp, const CostType* Lprev /* guards at [-1], [D] */, int minprev,
                      CostType* Lout, int* minout, int D, int P1, int P2)
{
    int delta = minprev + P2, mn = MAX_COST;
    for (int d = 0; d < D; d++) {
        int L = Cp[d] + imin((int)Lprev[d], imin(Lprev[d - 1] + P1, imin(Lprev[d + 1] + P1, delta))) - delta;
        Lout[d] = (CostType)L;
        mn = imin(mn, L);
    }
    *minout = mn;
}

static int compute_disparity_hh4(const PixType* img1, const PixType* img2, int width, int height, int cn,
                                 size_t step, DispType* disp1, const oracle_sgbm_params* params,
                                 const capture* cap)
{
    int minD = params->minDisparity, maxD = minD + params->numDisparities;
    int uniquenessRatio = params->uniquenessRatio >= 0 ? params->uniquenessRatio : 10;
    int disp12MaxDiff = params->disp12MaxDiff > 0 ? params->disp12MaxDiff : 1;
    int P1 = params->P1 > 0 ? params->P1 : 2;
    int P2 = imax(params->P2 > 0 ? params->P2 : 5, P1 + 1);
    int minX1 = imax(maxD, 0), maxX1 = width + imin(minD, 0);
    const int D = params->numDisparities;
    int width1 = maxX1 - minX1;
    int INVALID_DISP = minD - 1, INVALID_DISP_SCALED = INVALID_DISP * DISP_SCALE;
    int SADWindowSize = params->blockSize > 0 ? params->blockSize : 5;
    if (minX1 >= maxX1) {
        for (size_t i = 0; i < (size_t)width * height; i++) disp1[i] = (DispType)INVALID_DISP_SCALED;
        return 0;
    }
    if (width1 <= SADWindowSize / 2) return -2;

    size_t costWidth = (size_t)width1 * D, vol = costWidth * height;
    CostType* C = (CostType*)malloc(vol * sizeof(CostType));
    CostType* S = (CostType*)calloc(vol, sizeof(CostType));
    CostType* Lbuf = (CostType*)malloc((size_t)(D + 2) * 2 * sizeof(CostType));
    CostType* disp2cost = (CostType*)malloc((size_t)width * sizeof(CostType));
    DispType* disp2ptr = (DispType*)malloc((size_t)width * sizeof(DispType));
    DispType* scratch = (DispType*)malloc((size_t)width * height * sizeof(DispType));
    if (!C || !S || !Lbuf || !disp2cost || !disp2ptr || !scratch) {
        free(C); free(S); free(Lbuf); free(disp2cost); free(disp2ptr); free(scratch);
        return -1;
    }
    {   /* the cost volume of the row-incremental code path (mode 0 computes the same C) */
        oracle_sgbm_params q = *params;
        q.mode = 0;
        capture ccap = {C, NULL};
        int rc = compute_disparity_sgbm(img1, img2, width, height, cn, step, scratch, &q, &ccap);
        if (rc) { free(C); free(S); free(Lbuf); free(disp2cost); free(disp2ptr); free(scratch); return rc; }
    }
    if (cap && cap->C) memcpy(cap->C, C, vol * sizeof(CostType));
    CostType* La = Lbuf + 1;
    CostType* Lb = Lbuf + (D + 2) + 1;

    /* vertical stage */
    for (int x = 0; x < width1; x++) {
        for (int dir = 0; dir < 2; dir++) {
            CostType *prev = La, *cur = Lb;
            int minprev = 0;
            memset(prev, 0, (size_t)D * sizeof(CostType));
            for (int k = 0; k < height; k++) {
                int y = dir == 0 ? k : height - 1 - k;
                prev[-1] = prev[D] = MAX_COST;
                int mn;
                lr_update(C + (size_t)y * costWidth + (size_t)x * D, prev, minprev, cur, &mn, D, P1, P2);
                CostType* Sp = S + (size_t)y * costWidth + (size_t)x * D;
                for (int d = 0; d < D; d++) Sp[d] = sat16(Sp[d] + cur[d]);
                minprev = mn;
                CostType* t = prev; prev = cur; cur = t;
            }
        }
    }

    /* horizontal stage */
    for (int y = 0; y < height; y++) {
        DispType* disp1ptr = disp1 + (size_t)y * width;
        const CostType* Crow = C + (size_t)y * costWidth;
        CostType* Srow = S + (size_t)y * costWidth;
        int x, d;
        {   /* left to right */
            CostType *prev = La, *cur = Lb;
            int minprev = 0;
            memset(prev, 0, (size_t)D * sizeof(CostType));
            for (x = 0; x < width1; x++) {
                prev[-1] = prev[D] = MAX_COST;
                int mn;
                lr_update(Crow + (size_t)x * D, prev, minprev, cur, &mn, D, P1, P2);
                CostType* Sp = Srow + (size_t)x * D;
                for (d = 0; d < D; d++) Sp[d] = sat16(Sp[d] + cur[d]);
                minprev = mn;
                CostType* t = prev; prev = cur; cur = t;
            }
        }
        for (x = 0; x < width; x++) {
            disp1ptr[x] = disp2ptr[x] = (DispType)INVALID_DISP_SCALED;
            disp2cost[x] = MAX_COST;
        }
        {   /* right to left + winner-take-all */
            CostType *prev = La, *cur = Lb;
            int minprev = 0;
            memset(prev, 0, (size_t)D * sizeof(CostType));
            for (x = width1 - 1; x >= 0; x--) {
                prev[-1] = prev[D] = MAX_COST;
                int mn;
                lr_update(Crow + (size_t)x * D, prev, minprev, cur, &mn, D, P1, P2);
                CostType* Sp = Srow + (size_t)x * D;
                int minS = MAX_COST, bestDisp = -1;
                for (d = 0; d < D; d++) {
                    int Sval = Sp[d] = sat16(Sp[d] + cur[d]);
                    if (Sval < minS) { minS = Sval; bestDisp = d; }
                }
                minprev = mn;
                { CostType* t = prev; prev = cur; cur = t; }

                for (d = 0; d < D; d++)
                    if (Sp[d] * (100 - uniquenessRatio) < minS * 100 && abs(bestDisp - d) > 1) break;
                if (d < D) continue;
                d = bestDisp;
                int _x2 = x + minX1 - d - minD;
                if (disp2cost[_x2] > minS) {
                    disp2cost[_x2] = (CostType)minS;
                    disp2ptr[_x2] = (DispType)(d + minD);
                }
                if (0 < d && d < D - 1) {
                    int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                    d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
                } else
                    d *= DISP_SCALE;
                disp1ptr[x + minX1] = (DispType)(d + minD * DISP_SCALE);
            }
        }
        for (x = minX1; x < maxX1; x++) {
            int d1 = disp1ptr[x];
            if (d1 == INVALID_DISP_SCALED) continue;
            int _d = d1 >> DISP_SHIFT;
            int d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
            int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < width && disp2ptr[_x] >= minD && abs(disp2ptr[_x] - _d) > disp12MaxDiff &&
                0 <= x_ && x_ < width && disp2ptr[x_] >= minD && abs(disp2ptr[x_] - d_) > disp12MaxDiff)
                disp1ptr[x] = (DispType)INVALID_DISP_SCALED;
        }
    }
    if (cap && cap->S) memcpy(cap->S, S, vol * sizeof(CostType));
    free(C); free(S); free(Lbuf); free(disp2cost); free(disp2ptr); free(scratch);
    return 0;
}

/* ---- stereosgbm.cpp: computeDisparity3WaySGBM / SGBM3WayMainLoop (MODE_SGBM_3WAY = 2) -------------------------
 * Restated from recollection of OpenCV 4.x (no source at hand; every point I am not sure of is a U-flag in DESIGN.md):
 *   - the image is cut into `nstripes` = 4 horizontal stripes (fixed "to make the results fully reproducible"),
 *     stripe_sz = ceil(height / nstripes); a stripe starts `stripe_overlap` = blockSize/2 + 1 + ceil(0.1 * stripe_sz)
 *     rows early to warm up the vertical path and the box sums, and writes only its own rows;
 *   - per row: C (same BT cost + box sum + P2 as the other modes, but the vertical window is clamped at the stripe's
 *     first row), then a left-to-right pass that updates L_left (zero state at the row's left border) and, in place,
 *     L_top (zero state at the stripe's first row), then a right-to-left pass that updates L_right, sums the three
 *     and picks the winner; uniqueness (only if uniquenessRatio > 0), right-view map, sub-pixel, LR check as in SGBM;
 *   - SW2 = SH2 = blockSize > 0 ? blockSize/2 : 1 (not the "5 -> 2" of the other modes);
 *   - the winner-take-all of the CV_SIMD build is not "smallest d": see oracle_switches.way3_simd_lanes.           */
static void way3_winner(const CostType* tot, int D, int lanes, int* best_io, int* min_io)
{
    int best = *best_io, minc = SHRT_MAX; /* best_d survives from the previous pixel when nothing is below SHRT_MAX */
    int E = 0;
    if (lanes > 1) {
        E = (D % lanes == 0) ? D : lanes * ((D - 1) / lanes);
        if (E > 0) {
            int m = SHRT_MAX, pos = -1;
            for (int d = 0; d < E; d++) m = imin(m, tot[d]);
            for (int j = 0; j < lanes; j++) { /* per lane slot: the LAST d attaining the slot minimum */
                int sm = SHRT_MAX, sp = -1;
                for (int d = j; d < E; d += lanes)
                    if (tot[d] <= sm) { sm = tot[d]; sp = d; }
                if (sm == m && (pos < 0 || sp < pos)) pos = sp; /* smallest position among the slots holding it */
            }
            minc = m;
            best = pos;
        }
    }
    for (int d = E; d < D; d++) /* scalar build / scalar tail: strictly smaller wins */
        if (tot[d] < minc) { minc = tot[d]; best = d; }
    *best_io = best;
    *min_io = minc;
}

static int compute_disparity_3way(const PixType* img1, const PixType* img2, int width, int height, int cn, size_t step,
                                  DispType* disp1, const oracle_sgbm_params* params)
{
    const int DISP_SHIFT = 4, DISP_SCALE = 1 << DISP_SHIFT;
    int minD = params->minDisparity, maxD = minD + params->numDisparities, D = maxD - minD;
    int uniquenessRatio = params->uniquenessRatio >= 0 ? params->uniquenessRatio : 10;
    int disp12MaxDiff = params->disp12MaxDiff > 0 ? params->disp12MaxDiff : 1;
    int P1 = params->P1 > 0 ? params->P1 : 2, P2 = imax(params->P2 > 0 ? params->P2 : 5, P1 + 1);
    int minX1 = imax(maxD, 0), maxX1 = width + imin(minD, 0), width1 = maxX1 - minX1;
    int INVALID_DISP = minD - 1, INVALID_DISP_SCALED = INVALID_DISP * DISP_SCALE;
    int SW2 = params->blockSize > 0 ? params->blockSize / 2 : 1, SH2 = SW2;
    const int TAB_OFS = 256 * 4, TAB_SIZE = 256 + TAB_OFS * 2;
    PixType clipTab[256 + 256 * 4 * 2];
    int ftzero = imax(params->preFilterCap, 15) | 1;
    for (int k = 0; k < TAB_SIZE; k++) clipTab[k] = (PixType)(imin(imax(k - TAB_OFS, -ftzero), ftzero) + ftzero);

    if (minX1 >= maxX1) {
        for (size_t i = 0; i < (size_t)width * height; i++) disp1[i] = (DispType)INVALID_DISP_SCALED;
        return 0;
    }
    if (width1 <= SW2) return -2; /* the first box sum would read cost columns that were never computed */
    int nstripes = g_sw.way3_stripes > 0 ? g_sw.way3_stripes : 4;
    int stripe_sz = (height + nstripes - 1) / nstripes;
    int stripe_overlap = (params->blockSize / 2 + 1) + (int)ceil(0.1 * stripe_sz);
    /* OpenCV: "the stereo images cannot be very small" -- a later stripe whose warm-up would start above row 0 has
     * no well-defined row mapping there */
    for (int s = 1; s < nstripes; s++)
        if (s * stripe_sz < height && s * stripe_sz - stripe_overlap < 0) return -3;

    size_t costW = (size_t)width1 * D;
    int hsumRows = SH2 * 2 + 2;
    CostType* C = (CostType*)malloc(costW * sizeof(CostType));
    CostType* hsumBuf = (CostType*)malloc(costW * hsumRows * sizeof(CostType));
    CostType* pixDiff = (CostType*)malloc(costW * sizeof(CostType));
    CostType* hor = (CostType*)malloc((size_t)(width1 + 2) * D * sizeof(CostType));   /* L_left per column, then the total */
    CostType* vert = (CostType*)malloc((size_t)(width1 + 2) * D * sizeof(CostType));  /* L_top per column */
    CostType* vertMin = (CostType*)malloc((size_t)(width1 + 2) * sizeof(CostType));
    CostType* rightBuf = (CostType*)malloc((size_t)D * 2 * sizeof(CostType));
    CostType* tmpD = rightBuf + D;
    CostType* disp2cost = (CostType*)malloc((size_t)width * sizeof(CostType));
    DispType* disp2 = (DispType*)malloc((size_t)width * sizeof(DispType));
    DispType* disp_row = (DispType*)malloc((size_t)width * sizeof(DispType));
    PixType* tempBuf = (PixType*)malloc((size_t)width * (4 * cn + 2) + 64);
    if (!C || !hsumBuf || !pixDiff || !hor || !vert || !vertMin || !rightBuf || !disp2cost || !disp2 || !disp_row || !tempBuf) {
        free(C); free(hsumBuf); free(pixDiff); free(hor); free(vert); free(vertMin); free(rightBuf);
        free(disp2cost); free(disp2); free(disp_row); free(tempBuf);
        return -1;
    }
#define HS3(r) (hsumBuf + (size_t)((r) % hsumRows) * costW)

    for (int s = 0; s < nstripes; s++) {
        int src_start = imax(imin(s * stripe_sz - stripe_overlap, height), 0);
        int src_end = imin((s + 1) * stripe_sz, height);
        int first_out = s * stripe_sz; /* rows [first_out, src_end) belong to this stripe */
        for (size_t i = 0; i < costW; i++) C[i] = (CostType)P2; /* BufferSGBM3Way: curCostVolumeLine starts at P2 */
        memset(vert, 0, (size_t)(width1 + 2) * D * sizeof(CostType));
        memset(vertMin, 0, (size_t)(width1 + 2) * sizeof(CostType));
        int best_d = 0;

        for (int y = src_start; y < src_end; y++) {
            /* getRawMatchingCost(mem, y, src_start): the row buffer C is updated in place */
            int dy1 = (y == src_start) ? src_start : y + SH2, dy2 = (y == src_start) ? src_start + SH2 : dy1;
            for (int k = dy1; k <= dy2; k++) {
                CostType* hsumAdd = HS3(imin(k, height - 1));
                if (k < height) {
                    calc_pixel_cost_bt(img1, img2, step, width, height, cn, k, minD, maxD, pixDiff, tempBuf, clipTab + TAB_OFS);
                    for (int d = 0; d < D; d++) {
                        CostType h = c_mul(pixDiff[d], SW2 + 1);
                        for (int x = D; x <= SW2 * D; x += D) h = c_add(h, pixDiff[x + d]);
                        hsumAdd[d] = h;
                    }
                    for (int x = D; x < width1 * D; x += D) {
                        const CostType* pixAdd = pixDiff + imin(x + SW2 * D, (width1 - 1) * D);
                        const CostType* pixSub = pixDiff + imax(x - (SW2 + 1) * D, 0);
                        for (int d = 0; d < D; d++)
                            hsumAdd[x + d] = c_add(c_sub(hsumAdd[x - D + d], pixSub[d]), pixAdd[d]);
                    }
                }
                if (y > src_start) {
                    const CostType* hsumSub = HS3(imax(y - SH2 - 1, src_start));
                    /* same operation order as computeDisparitySGBM (it matters only past int16 saturation) */
                    for (size_t i = 0; i < costW; i++)
                        C[i] = (i < (size_t)D && k < height) ? c_sub(c_add(C[i], hsumAdd[i]), hsumSub[i])
                                                              : c_add(c_sub(C[i], hsumSub[i]), hsumAdd[i]);
                } else {
                    int scale = k == src_start ? SH2 + 1 : 1;
                    for (size_t i = 0; i < costW; i++) C[i] = c_add(C[i], c_mul(hsumAdd[i], scale));
                }
            }

            for (int x = 0; x < width; x++) {
                disp_row[x] = disp2[x] = (DispType)INVALID_DISP_SCALED;
                disp2cost[x] = SHRT_MAX;
            }
            /* forward pass: L_left (column x at hor[(x+1)*D], zero border at hor[0]) and L_top in place */
            memset(hor, 0, (size_t)D * sizeof(CostType));
            int leftMin = 0;
            for (int x = 0; x < width1; x++) {
                const CostType* Cp = C + (size_t)x * D;
                const CostType* lprev = hor + (size_t)x * D;
                CostType* left = hor + (size_t)(x + 1) * D;
                CostType* top = vert + (size_t)(x + 1) * D;
                int lP2 = leftMin + P2, tP2 = vertMin[x + 1] + P2, lmin = SHRT_MAX, tmin = SHRT_MAX;
                memcpy(tmpD, top, (size_t)D * sizeof(CostType));
                for (int d = 0; d < D; d++) {
                    int lm = d > 0 ? lprev[d - 1] : SHRT_MAX, lp = d < D - 1 ? lprev[d + 1] : SHRT_MAX;
                    int tm = d > 0 ? tmpD[d - 1] : SHRT_MAX, tp = d < D - 1 ? tmpD[d + 1] : SHRT_MAX;
                    left[d] = sat16(Cp[d] + imin(imin(lm + P1, lp + P1), imin((int)lprev[d], lP2)) - lP2);
                    top[d] = sat16(Cp[d] + imin(imin(tm + P1, tp + P1), imin((int)tmpD[d], tP2)) - tP2);
                    lmin = imin(lmin, left[d]);
                    tmin = imin(tmin, top[d]);
                }
                leftMin = lmin;
                vertMin[x + 1] = (CostType)tmin;
            }
            /* backward pass: L_right in place, total = right + left + top, winner */
            memset(rightBuf, 0, (size_t)D * sizeof(CostType));
            int rightMin = 0;
            for (int x = width1 - 1; x >= 0; x--) {
                const CostType* Cp = C + (size_t)x * D;
                CostType* tot = hor + (size_t)(x + 1) * D;
                const CostType* top = vert + (size_t)(x + 1) * D;
                int rP2 = rightMin + P2, rmin = SHRT_MAX, min_cost;
                memcpy(tmpD, rightBuf, (size_t)D * sizeof(CostType));
                for (int d = 0; d < D; d++) {
                    int rm = d > 0 ? tmpD[d - 1] : SHRT_MAX, rp = d < D - 1 ? tmpD[d + 1] : SHRT_MAX;
                    rightBuf[d] = sat16(Cp[d] + imin(imin(rm + P1, rp + P1), imin((int)tmpD[d], rP2)) - rP2);
                    rmin = imin(rmin, rightBuf[d]);
                    tot[d] = sat16(sat16(rightBuf[d] + tot[d]) + top[d]);
                }
                rightMin = rmin;
                way3_winner(tot, D, g_sw.way3_simd_lanes, &best_d, &min_cost);
                if (min_cost >= SHRT_MAX) continue; /* every total saturated: no winner */
                int d;
                if (uniquenessRatio > 0) {
                    for (d = 0; d < D; d++)
                        if (tot[d] * (100 - uniquenessRatio) < min_cost * 100 && abs(d - best_d) > 1) break;
                    if (d < D) continue;
                }
                d = best_d;
                int _x2 = x + minX1 - d - minD;
                if (_x2 >= 0 && _x2 < width && disp2cost[_x2] > min_cost) {
                    disp2cost[_x2] = (CostType)min_cost;
                    disp2[_x2] = (DispType)(d + minD);
                }
                if (0 < d && d < D - 1) {
                    int denom2 = imax(tot[d - 1] + tot[d + 1] - 2 * tot[d], 1);
                    d = d * DISP_SCALE + ((tot[d - 1] - tot[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
                } else
                    d *= DISP_SCALE;
                disp_row[x + minX1] = (DispType)(d + minD * DISP_SCALE);
            }
            for (int x = minX1; x < maxX1; x++) {
                int d1 = disp_row[x];
                if (d1 == INVALID_DISP_SCALED) continue;
                int _d = d1 >> DISP_SHIFT, d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
                int _x = x - _d, x_ = x - d_;
                if (0 <= _x && _x < width && disp2[_x] >= minD && abs(disp2[_x] - _d) > disp12MaxDiff &&
                    0 <= x_ && x_ < width && disp2[x_] >= minD && abs(disp2[x_] - d_) > disp12MaxDiff)
                    disp_row[x] = (DispType)INVALID_DISP_SCALED;
            }
            if (y >= first_out) memcpy(disp1 + (size_t)y * width, disp_row, (size_t)width * sizeof(DispType));
        }
    }
#undef HS3
    free(C); free(hsumBuf); free(pixDiff); free(hor); free(vert); free(vertMin); free(rightBuf);
    free(disp2cost); free(disp2); free(disp_row); free(tempBuf);
    return 0;
}

/* mode dispatch: 0 / 1 -> computeDisparitySGBM, 2 -> computeDisparity3WaySGBM, 3 -> computeDisparitySGBM_HH4 */
static int compute_disparity(const PixType* img1, const PixType* img2, int width, int height, int cn, size_t step,
                             DispType* disp1, const oracle_sgbm_params* params, const capture* cap)
{
    if (params->mode == 3) return compute_disparity_hh4(img1, img2, width, height, cn, step, disp1, params, cap);
    if (params->mode == 2) return compute_disparity_3way(img1, img2, width, height, cn, step, disp1, params);
    return compute_disparity_sgbm(img1, img2, width, height, cn, step, disp1, params, cap);
}

/* ---- median_blur.simd.hpp: medianBlur_SortNet, m == 3, int16, replicate border ------------- */
static inline void mm_op(int* a, int* b)
{
    int t = *a;
    *a = imin(*a, *b);
    *b = imax(*b, t);
}

void oracle_median3_s16(const int16_t* src, int16_t* dst, int width, int height)
{
    for (int i = 0; i < height; i++) {
        const int16_t* row0 = src + (size_t)imax(i - 1, 0) * width;
        const int16_t* row1 = src + (size_t)i * width;
        const int16_t* row2 = src + (size_t)imin(i + 1, height - 1) * width;
        for (int j = 0; j < width; j++) {
            int j0 = j >= 1 ? j - 1 : j;
            int j2 = j < width - 1 ? j + 1 : j;
            int p0 = row0[j0], p1 = row0[j], p2 = row0[j2];
            int p3 = row1[j0], p4 = row1[j], p5 = row1[j2];
            int p6 = row2[j0], p7 = row2[j], p8 = row2[j2];

            mm_op(&p1, &p2); mm_op(&p4, &p5); mm_op(&p7, &p8); mm_op(&p0, &p1);
            mm_op(&p3, &p4); mm_op(&p6, &p7); mm_op(&p1, &p2); mm_op(&p4, &p5);
            mm_op(&p7, &p8); mm_op(&p0, &p3); mm_op(&p5, &p8); mm_op(&p4, &p7);
            mm_op(&p3, &p6); mm_op(&p1, &p4); mm_op(&p2, &p5); mm_op(&p4, &p7);
            mm_op(&p4, &p2); mm_op(&p6, &p4); mm_op(&p4, &p2);
            dst[(size_t)i * width + j] = (int16_t)p4;
        }
    }
}

/* ---- stereosgbm.cpp: filterSpecklesImpl<short> ----------------------------------------------- */
void oracle_filter_speckles_s16(int16_t* img, int width, int height, int newVal, int maxSpeckleSize,
                                int maxDiff)
{
    size_t npixels = (size_t)width * height;
    int* labels = (int*)calloc(npixels, sizeof(int));
    int* wbuf = (int*)malloc(npixels * 2 * sizeof(int)); /* (x, y) pairs */
    uint8_t* rtype = (uint8_t*)malloc(npixels + 1);
    int curlabel = 0;

    for (int i = 0; i < height; i++) {
        int16_t* ds = img + (size_t)i * width;
        int* ls = labels + (size_t)width * i;

        for (int j = 0; j < width; j++) {
            if (ds[j] != newVal) /* not a bad disparity */
            {
                if (ls[j]) /* has a label, check for bad label */
                {
                    if (rtype[ls[j]]) /* small region, zero out disparity */
                        ds[j] = (int16_t)newVal;
                }
                /* no label, assign and propagate */
                else {
                    int* ws = wbuf;      /* initialize wavefront */
                    int px = j, py = i;  /* current pixel */
                    curlabel++;          /* next label */
                    int count = 0;       /* current region size */
                    ls[j] = curlabel;

                    /* wavefront propagation */
                    while (ws >= wbuf) /* wavefront not empty */
                    {
                        count++;
                        /* put neighbors onto wavefront */
                        int16_t* dpp = img + (size_t)py * width + px;
                        int dp = *dpp;
                        int* lpp = labels + (size_t)width * py + px;

                        if (py < height - 1 && !lpp[+width] && dpp[+width] != newVal && abs(dp - dpp[+width]) <= maxDiff) {
                            lpp[+width] = curlabel;
                            *ws++ = px; *ws++ = py + 1;
                        }
                        if (py > 0 && !lpp[-width] && dpp[-width] != newVal && abs(dp - dpp[-width]) <= maxDiff) {
                            lpp[-width] = curlabel;
                            *ws++ = px; *ws++ = py - 1;
                        }
                        if (px < width - 1 && !lpp[+1] && dpp[+1] != newVal && abs(dp - dpp[+1]) <= maxDiff) {
                            lpp[+1] = curlabel;
                            *ws++ = px + 1; *ws++ = py;
                        }
                        if (px > 0 && !lpp[-1] && dpp[-1] != newVal && abs(dp - dpp[-1]) <= maxDiff) {
                            lpp[-1] = curlabel;
                            *ws++ = px - 1; *ws++ = py;
                        }

                        /* pop most recent and propagate */
                        ws -= 2;
                        if (ws >= wbuf) { px = ws[0]; py = ws[1]; }
                    }

                    /* assign label type */
                    if (count <= maxSpeckleSize) /* speckle region */
                    {
                        rtype[ls[j]] = 1; /* small region label */
                        ds[j] = (int16_t)newVal;
                    } else
                        rtype[ls[j]] = 0; /* large region label */
                }
            }
        }
    }
    free(labels); free(wbuf); free(rtype);
}

/* ---- stereosgbm.cpp: StereoSGBMImpl::compute -------------------------------------------------- */
static int check_args(const oracle_sgbm_params* p, int width, int height, int cn)
{
    if (!p || width <= 0 || height <= 0 || (cn != 1 && cn != 3)) return -1;
    if (p->numDisparities <= 0) return -1;
    if (p->mode < 0 || p->mode > 3) return -1;
    return 0;
}

int oracle_sgbm_raw(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                    int width, int height, int cn, size_t step, int16_t* disp)
{
    if (check_args(p, width, height, cn)) return -1;
    return compute_disparity(left, right, width, height, cn, step, disp, p, NULL);
}

int oracle_sgbm_compute(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                        int width, int height, int cn, size_t step, int16_t* disp)
{
    if (check_args(p, width, height, cn)) return -1;
    size_t n = (size_t)width * height;
    int16_t* tmp = (int16_t*)malloc(n * sizeof(int16_t));
    if (!tmp) return -1;
    int rc = compute_disparity(left, right, width, height, cn, step, tmp, p, NULL);
    if (rc == 0) {
        /* medianBlur(disp, disp, 3): in place = out of place on a copy */
        oracle_median3_s16(tmp, disp, width, height);
        if (p->speckleWindowSize > 0)
            oracle_filter_speckles_s16(disp, width, height, (p->minDisparity - 1) * DISP_SCALE,
                                       p->speckleWindowSize, DISP_SCALE * p->speckleRange);
    }
    free(tmp);
    return rc;
}

int oracle_sgbm_compute_batch(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                              int width, int height, int cn, int n, int nthreads, int16_t* disp)
{
    if (check_args(p, width, height, cn) || n < 0) return -1;
    size_t img = (size_t)width * height * cn, dsz = (size_t)width * height;
    int rc = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        int r = oracle_sgbm_compute(p, left + img * i, right + img * i, width, height, cn,
                                    (size_t)width * cn, disp + dsz * i);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

int oracle_sgbm_cost_volume(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                            int width, int height, int cn, size_t step, int16_t* C)
{
    if (check_args(p, width, height, cn)) return -1;
    int16_t* tmp = (int16_t*)malloc((size_t)width * height * sizeof(int16_t));
    if (!tmp) return -1;
    capture cap = {C, NULL};
    int rc = compute_disparity(left, right, width, height, cn, step, tmp, p, &cap);
    free(tmp);
    return rc;
}

int oracle_sgbm_aggregated(const oracle_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                           int width, int height, int cn, size_t step, int16_t* S)
{
    if (check_args(p, width, height, cn)) return -1;
    int16_t* tmp = (int16_t*)malloc((size_t)width * height * sizeof(int16_t));
    if (!tmp) return -1;
    capture cap = {NULL, S};
    int rc = compute_disparity(left, right, width, height, cn, step, tmp, p, &cap);
    free(tmp);
    return rc;
}

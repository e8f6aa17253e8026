/*
 * remap_ref.c -- scalar CPU restatement of the cv2 geometry calls on the stereo-depth path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: restates, from the published
 * OpenCV 4.x sources (third-party dependency `opencv-contrib-python>=4.7.0.72`,
 * /root/reference/requirements.txt:2; not vendored, not importable here):
 *   opencv/modules/imgproc/src/imgwarp.cpp: interpolateLanczos4, initInterTab1D/2D, RemapInvoker,
 *       remapNearest, remapBilinear, remapLanczos4 (8-bit fixed point, BORDER_CONSTANT 0)
 *   opencv/modules/calib3d/src/undistort.dispatch.cpp: initUndistortRectifyMap, undistort
 * Reference call sites:
 *   /root/reference/calibrating/stereo_camera.py:159-165  initUndistortRectifyMap(..., CV_32FC1)
 *   /root/reference/calibrating/stereo_camera.py:217-228  remap(img, mapx, mapy, INTER_LANCZOS4)
 *   /root/reference/calibrating/stereo_camera.py:430-431  undistort(img1, K, D)
 *   /root/reference/calibrating/utils.py:184-191,199      initUndistortRectifyMap + remap(NEAREST)
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int oracle_sw_lanczos_group(void);

enum {
    INTER_BITS = 5,
    INTER_TAB_SIZE = 1 << INTER_BITS,
    INTER_TAB_SIZE2 = INTER_TAB_SIZE * INTER_TAB_SIZE,
    INTER_REMAP_COEF_BITS = 15,
    INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS
};

/* cvRound: round half to even (lrint under the default rounding mode) */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* imgwarp.cpp: interpolateLanczos4 */
static void interpolate_lanczos4(float x, float* coeffs)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45},
                                   {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    const double CV_PI = 3.1415926535897932384626433832795;
    float sum = 0;
    double y0 = -(x + 3) * CV_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        float y0_ = (x + 3 - i);
        if (fabsf(y0_) >= 1e-6f) {
            double y = -y0_ * CV_PI * 0.25;
            coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else {
            /* exact hit: this tap takes all the weight after normalisation */
            coeffs[i] = 1e30f;
        }
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

static void interpolate_linear(float x, float* coeffs)
{
    coeffs[0] = 1.f - x;
    coeffs[1] = x;
}

/* imgwarp.cpp: initInterTab2D(method, fixpt=true) -> itab[INTER_TAB_SIZE2][ksize*ksize] */
static void init_inter_tab_2d(int ksize, int16_t* itab)
{
    float tab1d[8 * INTER_TAB_SIZE];
    float scale = 1.f / INTER_TAB_SIZE;
    for (int i = 0; i < INTER_TAB_SIZE; i++) {
        if (ksize == 8)
            interpolate_lanczos4(i * scale, tab1d + i * 8);
        else
            interpolate_linear(i * scale, tab1d + i * 2);
    }
    /* OpenCV scans k1,k2 in [ksize/2, ksize/2+2) relative to the entry; for ksize 2 that window
       reaches past the 2x2 entry into the following entries of the (static, zero-initialised) table,
       which are not filled yet when the entry is processed.  Restated literally: candidates are read
       from the whole table (zero where not yet written, skipped past its end).  It only fires at
       phase (0,0), where saturate_cast<short>(32768) = 32767 leaves the sum one short. */
    int glo = ksize == 8 ? oracle_sw_lanczos_group() : ksize / 2;
    int16_t* const tab0 = itab;
    const ptrdiff_t total = (ptrdiff_t)INTER_TAB_SIZE2 * ksize * ksize;
    memset(itab, 0, (size_t)total * sizeof(int16_t));
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++, itab += ksize * ksize) {
            int isum = 0;
            for (int k1 = 0; k1 < ksize; k1++) {
                float vy = tab1d[i * ksize + k1];
                for (int k2 = 0; k2 < ksize; k2++) {
                    float v = vy * tab1d[j * ksize + k2];
                    isum += itab[k1 * ksize + k2] = sat_short(cv_round_f(v * INTER_REMAP_COEF_SCALE));
                }
            }
            if (isum != INTER_REMAP_COEF_SCALE) {
                int diff = isum - INTER_REMAP_COEF_SCALE;
                int Mk1 = glo, Mk2 = glo, mk1 = glo, mk2 = glo;
                const ptrdiff_t room = total - (itab - tab0);
                for (int k1 = glo; k1 < glo + 2; k1++)
                    for (int k2 = glo; k2 < glo + 2; k2++) {
                        if (k1 * ksize + k2 >= room) continue;
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2])
                            mk1 = k1, mk2 = k2;
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2])
                            Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0)
                    itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else
                    itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
}

void oracle_lanczos4_itab(int16_t* tab) { init_inter_tab_2d(8, tab); }
void oracle_bilinear_itab(int16_t* tab) { init_inter_tab_2d(2, tab); }

/* remapLanczos4 / remapBilinear<FixedPtCast<int,uchar,15>> with BORDER_CONSTANT(0), one pixel.
 * (sx, sy) = integer source position of tap (0,0); w = ksize*ksize weights.                      */
static void remap_fixed_pixel(const uint8_t* src, int sw, int sh, int cn, int ksize, int sx, int sy,
                              const int16_t* w, uint8_t* D)
{
    if (sx >= sw || sx + ksize <= 0 || sy >= sh || sy + ksize <= 0) {
        for (int k = 0; k < cn; k++) D[k] = 0;
        return;
    }
    for (int k = 0; k < cn; k++) {
        int sum = 0;
        for (int r = 0; r < ksize; r++) {
            int yy = sy + r;
            if (yy < 0 || yy >= sh) continue;
            for (int c = 0; c < ksize; c++) {
                int xx = sx + c;
                if (xx < 0 || xx >= sw) continue;
                sum += src[((size_t)yy * sw + xx) * cn + k] * w[r * ksize + c];
            }
        }
        D[k] = sat_u8((sum + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS);
    }
}

int oracle_remap_u8(const uint8_t* src, int sw, int sh, int cn, const float* mapx, const float* mapy,
                    uint8_t* dst, int dw, int dh, int interp)
{
    if (interp == 0) {
        for (size_t i = 0; i < (size_t)dw * dh; i++) {
            int sx = sat_short(cv_round_f(mapx[i])), sy = sat_short(cv_round_f(mapy[i]));
            for (int k = 0; k < cn; k++)
                dst[i * cn + k] = ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh)
                                      ? src[((size_t)sy * sw + sx) * cn + k] : 0;
        }
        return 0;
    }
    int ksize = interp == 4 ? 8 : interp == 1 ? 2 : 0;
    if (!ksize) return -1;
    int16_t* itab = (int16_t*)malloc((size_t)INTER_TAB_SIZE2 * ksize * ksize * sizeof(int16_t));
    if (!itab) return -1;
    init_inter_tab_2d(ksize, itab);
    int ofs = ksize / 2 - 1; /* 3 for Lanczos4, 0 for bilinear */
    for (size_t i = 0; i < (size_t)dw * dh; i++) {
        /* RemapInvoker: float map * INTER_TAB_SIZE in float, cvRound, split */
        int sx = cv_round_f(mapx[i] * INTER_TAB_SIZE);
        int sy = cv_round_f(mapy[i] * INTER_TAB_SIZE);
        int a = (sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1));
        int ix = sat_short(sx >> INTER_BITS), iy = sat_short(sy >> INTER_BITS);
        remap_fixed_pixel(src, sw, sh, cn, ksize, ix - ofs, iy - ofs, itab + (size_t)a * ksize * ksize,
                          dst + i * cn);
    }
    free(itab);
    return 0;
}

void oracle_remap_nearest_f64(const double* src, int sw, int sh, const float* mapx,
                              const float* mapy, double* dst, int dw, int dh)
{
    for (size_t i = 0; i < (size_t)dw * dh; i++) {
        int sx = sat_short(cv_round_f(mapx[i])), sy = sat_short(cv_round_f(mapy[i]));
        dst[i] = ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh)
                     ? src[(size_t)sy * sw + sx] : 0.0;
    }
}

/* 3x3 inverse via LU with partial pivoting (cv::Mat::inv(DECOMP_LU) on a 3x3 double uses the
 * closed-form adjugate for n<=3; both are restated: closed form is what OpenCV takes for 3x3) */
static void inv3(const double m[9], double out[9])
{
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    d = d != 0. ? 1. / d : 0.;
    double t[9];
    t[0] = (m[4] * m[8] - m[5] * m[7]) * d;
    t[1] = (m[2] * m[7] - m[1] * m[8]) * d;
    t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    t[3] = (m[5] * m[6] - m[3] * m[8]) * d;
    t[4] = (m[0] * m[8] - m[2] * m[6]) * d;
    t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    t[6] = (m[3] * m[7] - m[4] * m[6]) * d;
    t[7] = (m[1] * m[6] - m[0] * m[7]) * d;
    t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    memcpy(out, t, sizeof(t));
}

static void matmul3(const double a[9], const double b[9], double c[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
            c[i * 3 + j] = s;
        }
}

typedef struct dist_t { double k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4; } dist_t;

static dist_t load_dist(const double* dist, int ndist)
{
    double v[14] = {0};
    for (int i = 0; i < ndist && i < 14; i++) v[i] = dist ? dist[i] : 0.;
    dist_t d = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11]};
    return d; /* tauX/tauY (tilted sensor) are 0 on this path: reference D is (1,5), camera.py:442 */
}

/* one row of initUndistortRectifyMapComputer (scalar form: X/Y/W accumulate per column) */
static void undistort_row(const double ir[9], double fx, double fy, double u0, double v0,
                          const dist_t* k, int i, int w, double* u, double* v)
{
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        double ww = 1. / _w, x = _x * ww, y = _y * ww;
        double x2 = x * x, y2 = y * y;
        double r2 = x2 + y2, _2xy = 2 * x * y;
        double kr = (1 + ((k->k3 * r2 + k->k2) * r2 + k->k1) * r2) /
                    (1 + ((k->k6 * r2 + k->k5) * r2 + k->k4) * r2);
        double xd = (x * kr + k->p1 * _2xy + k->p2 * (r2 + 2 * x2) + k->s1 * r2 + k->s2 * r2 * r2);
        double yd = (y * kr + k->p1 * (r2 + 2 * y2) + k->p2 * _2xy + k->s3 * r2 + k->s4 * r2 * r2);
        /* matTilt = I  ->  invProj = 1 */
        u[j] = fx * xd + u0;
        v[j] = fy * yd + v0;
    }
}

void oracle_init_undistort_rectify_map(const double A[9], const double* dist, int ndist,
                                       const double* R, const double Anew[9], int w, int h,
                                       float* mapx, float* mapy)
{
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double ArR[9], ir[9];
    matmul3(Anew, R ? R : I3, ArR);
    inv3(ArR, ir);
    dist_t k = load_dist(dist, ndist);
    double* u = (double*)malloc(sizeof(double) * w * 2);
    double* v = u + w;
    for (int i = 0; i < h; i++) {
        undistort_row(ir, A[0], A[4], A[2], A[5], &k, i, w, u, v);
        for (int j = 0; j < w; j++) {
            mapx[(size_t)i * w + j] = (float)u[j];
            mapy[(size_t)i * w + j] = (float)v[j];
        }
    }
    free(u);
}

/* undistort.dispatch.cpp: cv::undistort -- stripes of rows, CV_16SC2 + CV_16UC1 maps computed
 * with the stripe's row offset folded into Ar(1,2), then remap(INTER_LINEAR, BORDER_CONSTANT). */
void oracle_undistort_u8(const uint8_t* src, int w, int h, int cn, const double K[9],
                         const double* dist, int ndist, uint8_t* dst)
{
    int stripe_size0 = (1 << 12) / (w > 1 ? w : 1);
    if (stripe_size0 < 1) stripe_size0 = 1;
    if (stripe_size0 > h) stripe_size0 = h;
    dist_t k = load_dist(dist, ndist);
    int16_t itab[INTER_TAB_SIZE2 * 4];
    init_inter_tab_2d(2, itab);
    double* u = (double*)malloc(sizeof(double) * w * 2);
    double* v = u + w;
    double Ar[9], ir[9];
    memcpy(Ar, K, sizeof(Ar));
    double v0 = Ar[5];
    for (int y = 0; y < h; y += stripe_size0) {
        int stripe = stripe_size0 < h - y ? stripe_size0 : h - y;
        Ar[5] = v0 - y;
        inv3(Ar, ir); /* R = I */
        for (int i = 0; i < stripe; i++) {
            undistort_row(ir, K[0], K[4], K[2], K[5], &k, i, w, u, v);
            for (int j = 0; j < w; j++) {
                int iu = cv_round_d(u[j] * INTER_TAB_SIZE);
                int iv = cv_round_d(v[j] * INTER_TAB_SIZE);
                int ix = (short)(iu >> INTER_BITS), iy = (short)(iv >> INTER_BITS);
                int a = (iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1));
                remap_fixed_pixel(src, w, h, cn, 2, ix, iy, itab + a * 4,
                                  dst + ((size_t)(y + i) * w + j) * cn);
            }
        }
    }
    free(u);
}

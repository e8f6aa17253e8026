// tables.hip -- the rig's lookup tables built on the GPU: cv2.initUndistortRectifyMap
// (CV_32FC1 maps, stereo_camera.py:159-165 and utils.py:184-191) with the rectify valid mask
// (stereo_camera.py:167-176) fused, and the CV_16SC2 + CV_16UC1 maps cv2.undistort builds internally
// (stereo_camera.py:430-431).  Bit-identical to the host construction (geometry.py / camd_undistort_maps_host)
// and to the oracle: float64, no contraction (-ffp-contract=off), correctly rounded division.
//
// OpenCV accumulates X, Y, W along a row by repeated addition (_x += ir[0] ...).  That recurrence is the
// only sequential part: a workgroup owns one row, three of its lanes run the three chains for a chunk of
// columns into LDS (a few microseconds; all rows run in parallel), then all 256 lanes do the per-pixel
// distortion arithmetic from LDS.
#include "common.hpp"

namespace camd {

struct DistK { double k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4; };

struct TableArgs {
    double A[9];     // camera matrix of the SOURCE image (fx, fy, cx, cy used)
    double New[9];   // new camera matrix * R (its inverse maps destination pixels to rays)
    DistK k;
    int w, h;        // destination size
    int src_w, src_h;  // valid-mask bounds (mask != nullptr)
    int stripe;      // fixed-point variant: rows per stripe (cv2.undistort folds the stripe offset into cy)
};

__host__ __device__ inline void inv3_rm(const double* m, double* o)
{
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    d = d != 0. ? 1. / d : 0.;
    double t[9] = {(m[4] * m[8] - m[5] * m[7]) * d, (m[2] * m[7] - m[1] * m[8]) * d,
                   (m[1] * m[5] - m[2] * m[4]) * d, (m[5] * m[6] - m[3] * m[8]) * d,
                   (m[0] * m[8] - m[2] * m[6]) * d, (m[2] * m[3] - m[0] * m[5]) * d,
                   (m[3] * m[7] - m[4] * m[6]) * d, (m[1] * m[6] - m[0] * m[7]) * d,
                   (m[0] * m[4] - m[1] * m[3]) * d};
    for (int i = 0; i < 9; i++) o[i] = t[i];
}

static constexpr int TAB_CHUNK = 1024;  // columns per chunk: 3 x 8 KB of LDS

// FIXED = false: float maps (+ optional mask).  FIXED = true: int16 (x, y) + uint16 phase maps, row stripes.
template <bool FIXED>
__global__ __launch_bounds__(256) void k_undistort_rectify_map(TableArgs a, float* __restrict__ mapx,
                                                               float* __restrict__ mapy, uint8_t* __restrict__ mask,
                                                               int16_t* __restrict__ mapxy, uint16_t* __restrict__ mapa)
{
    __shared__ double sX[TAB_CHUNK], sY[TAB_CHUNK], sW[TAB_CHUNK];
    __shared__ double carry[3];
    const int row = blockIdx.x;
    double ir[9];
    int i = row;
    if (FIXED) {
        // stripe y0 = row - row % stripe: Ar = K with cy - y0, R = I
        const int y0 = row - row % a.stripe;
        double Ar[9];
        for (int q = 0; q < 9; q++) Ar[q] = a.New[q];
        Ar[5] = a.New[5] - y0;
        inv3_rm(Ar, ir);
        i = row - y0;
    } else {
        inv3_rm(a.New, ir);
    }
    const double fx = a.A[0], fy = a.A[4], u0 = a.A[2], v0 = a.A[5];
    const DistK k = a.k;
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        carry[c] = i * ir[3 * c + 1] + ir[3 * c + 2];
    }
    for (int j0 = 0; j0 < a.w; j0 += TAB_CHUNK) {
        const int n = min(TAB_CHUNK, a.w - j0);
        if (threadIdx.x < 3) {
            const int c = threadIdx.x;
            double* dst = c == 0 ? sX : (c == 1 ? sY : sW);
            const double step = ir[3 * c];
            double acc = carry[c];
            for (int j = 0; j < n; j++) {
                dst[j] = acc;
                acc += step;
            }
            carry[c] = acc;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256) {
            const double _x = sX[j], _y = sY[j], _w = sW[j];
            const double ww = 1. / _w, x = _x * ww, y = _y * ww;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k.k3 * r2 + k.k2) * r2 + k.k1) * r2) / (1 + ((k.k6 * r2 + k.k5) * r2 + k.k4) * r2);
            const double xd = (x * kr + k.p1 * _2xy + k.p2 * (r2 + 2 * x2) + k.s1 * r2 + k.s2 * r2 * r2);
            const double yd = (y * kr + k.p1 * (r2 + 2 * y2) + k.p2 * _2xy + k.s3 * r2 + k.s4 * r2 * r2);
            const double u = fx * xd + u0, v = fy * yd + v0;
            const size_t o = (size_t)row * a.w + j0 + j;
            if (FIXED) {
                const int iu = (int)rint(u * 32), iv = (int)rint(v * 32);  // cvRound(u * INTER_TAB_SIZE)
                mapxy[o * 2] = (int16_t)(iu >> 5);
                mapxy[o * 2 + 1] = (int16_t)(iv >> 5);
                mapa[o] = (uint16_t)((iv & 31) * 32 + (iu & 31));
            } else {
                const float fu = (float)u, fv = (float)v;
                mapx[o] = fu;
                mapy[o] = fv;
                if (mask)  // stereo_camera.py:167-176: (-0.5 < mapx) & (mapx < w - 0.5) & (-0.5 < mapy) & (mapy < h - 0.5)
                    mask[o] = (-0.5f < fu && fu < (float)a.src_w - 0.5f && -0.5f < fv && fv < (float)a.src_h - 0.5f) ? 1 : 0;
            }
        }
        __syncthreads();
    }
}

static int fill_args(TableArgs* a, const double A[9], const double* dist, int ndist, int w, int h, const char* who)
{
    if (!A || w <= 0 || h <= 0 || ndist < 0 || ndist > 14 || (ndist > 0 && !dist)) {
        set_error("%s: bad arguments", who);
        return CAMD_ERR_BAD_ARG;
    }
    double dv[14] = {0};
    for (int i = 0; i < ndist; i++) dv[i] = dist[i];
    if (dv[12] != 0. || dv[13] != 0.) {
        set_error("%s: tilted-sensor distortion (tauX, tauY) not implemented", who);
        return CAMD_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < 9; i++) a->A[i] = A[i];
    a->k = {dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7], dv[8], dv[9], dv[10], dv[11]};
    a->w = w;
    a->h = h;
    a->src_w = a->src_h = 0;
    a->stripe = 1;
    return CAMD_OK;
}

}  // namespace camd

using namespace camd;

extern "C" {

int camd_init_undistort_rectify_map(const double A[9], const double* dist, int ndist, const double* R,
                                    const double Anew[9], int w, int h, float* mapx, float* mapy,
                                    uint8_t* valid_mask, int src_w, int src_h, void* stream)
{
    TableArgs a;
    int rc = fill_args(&a, A, dist, ndist, w, h, "camd_init_undistort_rectify_map");
    if (rc != CAMD_OK) return rc;
    if (!Anew || !mapx || !mapy) { set_error("camd_init_undistort_rectify_map: NULL argument"); return CAMD_ERR_BAD_ARG; }
    rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double* Rm = R ? R : I3;
    for (int i = 0; i < 3; i++)  // Anew * R, the accumulation order of a plain triple loop (s = 0; s += a*b)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int q = 0; q < 3; q++) s += Anew[i * 3 + q] * Rm[q * 3 + j];
            a.New[i * 3 + j] = s;
        }
    a.src_w = src_w;
    a.src_h = src_h;
    hipLaunchKernelGGL((k_undistort_rectify_map<false>), dim3(h), dim3(256), 0, (hipStream_t)stream, a, mapx, mapy,
                       valid_mask, (int16_t*)nullptr, (uint16_t*)nullptr);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_undistort_maps(const double K[9], const double* dist, int ndist, int w, int h, int16_t* mapxy,
                        uint16_t* mapa, void* stream)
{
    TableArgs a;
    int rc = fill_args(&a, K, dist, ndist, w, h, "camd_undistort_maps");
    if (rc != CAMD_OK) return rc;
    if (!mapxy || !mapa) { set_error("camd_undistort_maps: NULL argument"); return CAMD_ERR_BAD_ARG; }
    rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    for (int i = 0; i < 9; i++) a.New[i] = K[i];
    int stripe0 = (1 << 12) / (w > 1 ? w : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > h) stripe0 = h;
    a.stripe = stripe0;
    hipLaunchKernelGGL((k_undistort_rectify_map<true>), dim3(h), dim3(256), 0, (hipStream_t)stream, a, (float*)nullptr,
                       (float*)nullptr, (uint8_t*)nullptr, mapxy, mapa);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

}  // extern "C"

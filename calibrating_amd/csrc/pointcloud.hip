// pointcloud.hip -- the depth post-ops that follow get_depth in the reference's demos:
//   utils.depth_to_point_cloud   (utils.py:213-246)   depth (+ optional INTER_NEAREST upsampling) -> N x 3 points
//   utils.apply_T_to_point_cloud (utils.py:152-161)
//   utils.point_cloud_to_depth   (utils.py:249-318)   projection + "far first, near last" overwrite = z-buffer
//   Cam.project_cam2_depth       (camera.py:298-309)  the three above composed; here ONE fused scatter kernel
// float64 throughout like the reference's NumPy.  Matrix products are evaluated left to right without
// contraction; NumPy's BLAS may order / fuse them differently, so parity with the oracle is to ~1 ulp on the
// points and exact on the z-buffer except where a projection lands within rounding error of x.5.
#include "common.hpp"

namespace camd {

struct PcGrid {
    int w, h;        // source depth
    int gw, gh;      // sampling grid (= w, h when rate == 1)
    double ifx, ify; // cv2.resize(INTER_NEAREST): sx = min(floor(x * ifx), w - 1)
    double rate;
};

__device__ __forceinline__ double pc_sample(const double* __restrict__ depth, const PcGrid& g, int x, int y)
{
    int sx = x, sy = y;
    if (g.gw != g.w || g.gh != g.h) {
        sx = min((int)floor(x * g.ifx), g.w - 1);
        sy = min((int)floor(y * g.ify), g.h - 1);
    }
    return depth[(size_t)sy * g.w + sx];
}

// order-preserving map double -> u64 (total order of the reals; -0.0 < +0.0)
// Matrix products the way NumPy's matmul rounds them: the reference's (K^-1 @ P.T).T, (T @ P4.T).T and P @ K.T are BLAS
// dgemm calls whose x86-64 kernels accumulate the k terms in order with fused multiply-adds, starting from the plain
// first product.  Measured against the reference's own run (tests/golden/reference_plumbing.npz) and against NumPy on
// every shape involved: this chain reproduces the bits, individually rounded products and sums differ in 1 of 4 values.
__device__ __forceinline__ double dot3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return __fma_rn(a2, b2, __fma_rn(a1, b1, __dmul_rn(a0, b0)));
}
__device__ __forceinline__ double dot4(double a0, double a1, double a2, double a3, double b0, double b1, double b2, double b3)
{
    return __fma_rn(a3, b3, __fma_rn(a2, b2, __fma_rn(a1, b1, __dmul_rn(a0, b0))));
}

__device__ __forceinline__ unsigned long long zkey(double z)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(z);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double zkey_inv(unsigned long long k)
{
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
static constexpr unsigned long long ZKEY_EMPTY = 0xffffffffffffffffull;

__global__ __launch_bounds__(256) void k_pc_count(const double* __restrict__ depth, PcGrid g, uint32_t* __restrict__ rowcount)
{
    __shared__ uint32_t part[4];
    const int y = blockIdx.x;
    uint32_t c = 0;
    for (int x = threadIdx.x; x < g.gw; x += 256) c += pc_sample(depth, g, x, y) != 0.0 ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) rowcount[y] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the row counts (one workgroup; rows <= a few thousand)
__global__ __launch_bounds__(256) void k_pc_scan(const uint32_t* __restrict__ rowcount, int n,
                                                 unsigned long long* __restrict__ rowoff,
                                                 unsigned long long* __restrict__ total)
{
    __shared__ unsigned long long carry;
    __shared__ unsigned long long wsum[4];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        unsigned long long v = i < n ? rowcount[i] : 0ull, incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            unsigned long long t = __shfl_up(incl, o);
            if ((threadIdx.x & 63) >= o) incl += t;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before += wsum[k];
        if (i < n) rowoff[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_pc_emit(const double* __restrict__ depth, PcGrid g,
                                                 const unsigned long long* __restrict__ rowoff, double Ki0, double Ki1,
                                                 double Ki2, double Ki3, double Ki4, double Ki5, double Ki6, double Ki7,
                                                 double Ki8, double* __restrict__ points, double* __restrict__ uv,
                                                 size_t capacity)
{
    __shared__ uint32_t wcnt[4];
    __shared__ unsigned long long run;
    const int y = blockIdx.x;
    if (threadIdx.x == 0) run = rowoff[y];
    __syncthreads();
    for (int base = 0; base < g.gw; base += 256) {
        const int x = base + threadIdx.x;
        const double z = x < g.gw ? pc_sample(depth, g, x, y) : 0.0;
        const bool nz = z != 0.0;
        const unsigned long long bal = __ballot(nz);
        const uint32_t below = __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull));
        if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(bal);
        __syncthreads();
        unsigned long long pos = run + below;
        for (int k = 0; k < (int)(threadIdx.x >> 6); k++) pos += wcnt[k];
        if (nz && pos < capacity) {
            // utils.py:225-246: us, vs are grid indices (/ interpolation_rate when upsampled)
            const double u = g.rate == 1.0 ? (double)x : (double)x / g.rate;
            const double v = g.rate == 1.0 ? (double)y : (double)y / g.rate;
            const double p0 = u * z, p1 = v * z, p2 = 1.0 * z;
            points[pos * 3 + 0] = dot3(Ki0, Ki1, Ki2, p0, p1, p2);
            points[pos * 3 + 1] = dot3(Ki3, Ki4, Ki5, p0, p1, p2);
            points[pos * 3 + 2] = dot3(Ki6, Ki7, Ki8, p0, p1, p2);
            if (uv) { uv[pos * 2] = u; uv[pos * 2 + 1] = v; }
        }
        __syncthreads();
        if (threadIdx.x == 0) run += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_fill_u64(unsigned long long* p, size_t n, unsigned long long v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Mat34 { double m[12]; };
struct Mat33 { double m[9]; };

// project one camera-space point with K (utils.py:286-288, 311-316) and keep the nearest per pixel
__device__ __forceinline__ void zbuffer_point(double X, double Y, double Z, const Mat33& K, int w, int h,
                                              unsigned long long* __restrict__ keys)
{
    const double xs = dot3(X, Y, Z, K.m[0], K.m[1], K.m[2]);
    const double ys = dot3(X, Y, Z, K.m[3], K.m[4], K.m[5]);
    const double zs = dot3(X, Y, Z, K.m[6], K.m[7], K.m[8]);
    const double u = xs / zs, v = ys / zs;
    const double ru = rint(u), rv = rint(v);  // np.round: half to even
    if (!(ru >= 0.0 && ru < (double)w && rv >= 0.0 && rv < (double)h)) return;  // also drops NaN / inf
    atomicMin(keys + (size_t)(int)rv * w + (int)ru, zkey(zs));
}

__global__ __launch_bounds__(256) void k_pc_scatter(const double* __restrict__ points, size_t n, int stride, Mat33 K, int w,
                                                    int h, unsigned long long* __restrict__ keys)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    zbuffer_point(points[i * stride], points[i * stride + 1], points[i * stride + 2], K, w, h, keys);
}

__global__ __launch_bounds__(256) void k_pc_resolve(const unsigned long long* __restrict__ keys, size_t n, double bg,
                                                    double* __restrict__ depth)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) depth[i] = keys[i] == ZKEY_EMPTY ? bg : zkey_inv(keys[i]);
}

__global__ __launch_bounds__(256) void k_apply_T(const double* __restrict__ src, size_t n, Mat34 T, double* __restrict__ dst)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = src[i * 3], y = src[i * 3 + 1], z = src[i * 3 + 2];
    dst[i * 3 + 0] = dot4(T.m[0], T.m[1], T.m[2], T.m[3], x, y, z, 1.0);
    dst[i * 3 + 1] = dot4(T.m[4], T.m[5], T.m[6], T.m[7], x, y, z, 1.0);
    dst[i * 3 + 2] = dot4(T.m[8], T.m[9], T.m[10], T.m[11], x, y, z, 1.0);
}

// Cam.project_cam2_depth fused: depth2 grid cell -> point (K2^-1) -> T -> K1 projection -> z-buffer
__global__ __launch_bounds__(256) void k_project_depth(const double* __restrict__ depth2, PcGrid g, Mat33 K2inv, Mat34 T,
                                                       Mat33 K1, int w1, int h1, unsigned long long* __restrict__ keys)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.gw) return;
    const double z = pc_sample(depth2, g, x, y);
    if (z == 0.0) return;
    const double u = g.rate == 1.0 ? (double)x : (double)x / g.rate;
    const double v = g.rate == 1.0 ? (double)y : (double)y / g.rate;
    const double p0 = u * z, p1 = v * z, p2 = 1.0 * z;
    const double X = dot3(K2inv.m[0], K2inv.m[1], K2inv.m[2], p0, p1, p2);
    const double Y = dot3(K2inv.m[3], K2inv.m[4], K2inv.m[5], p0, p1, p2);
    const double Z = dot3(K2inv.m[6], K2inv.m[7], K2inv.m[8], p0, p1, p2);
    const double X1 = dot4(T.m[0], T.m[1], T.m[2], T.m[3], X, Y, Z, 1.0);
    const double Y1 = dot4(T.m[4], T.m[5], T.m[6], T.m[7], X, Y, Z, 1.0);
    const double Z1 = dot4(T.m[8], T.m[9], T.m[10], T.m[11], X, Y, Z, 1.0);
    zbuffer_point(X1, Y1, Z1, K1, w1, h1, keys);
}

static int make_grid(PcGrid* g, int w, int h, double rate, const char* who)
{
    if (w <= 0 || h <= 0 || !(rate > 0.0)) { set_error("%s: bad size / interpolation rate", who); return CAMD_ERR_BAD_ARG; }
    g->w = w; g->h = h; g->rate = rate;
    if (rate == 1.0) {
        g->gw = w; g->gh = h; g->ifx = g->ify = 1.0;
    } else {
        // utils.py:231: y_, x_ = int(round(y * rate)), int(round(x * rate))  (Python round: half to even)
        g->gw = (int)nearbyint(w * rate);
        g->gh = (int)nearbyint(h * rate);
        if (g->gw <= 0 || g->gh <= 0) { set_error("%s: empty sampling grid", who); return CAMD_ERR_BAD_ARG; }
        g->ifx = 1.0 / ((double)g->gw / w);   // cv2.resize: inv_scale = dsize / ssize, ifx = 1 / inv_scale
        g->ify = 1.0 / ((double)g->gh / h);
    }
    return CAMD_OK;
}

}  // namespace camd

using namespace camd;

extern "C" {

int camd_point_cloud_grid(int w, int h, double rate, int* grid_w, int* grid_h)
{
    PcGrid g;
    int rc = make_grid(&g, w, h, rate, "camd_point_cloud_grid");
    if (rc != CAMD_OK) return rc;
    if (grid_w) *grid_w = g.gw;
    if (grid_h) *grid_h = g.gh;
    return CAMD_OK;
}

size_t camd_point_cloud_workspace_bytes(int w, int h, double rate)
{
    PcGrid g;
    if (make_grid(&g, w, h, rate, "camd_point_cloud_workspace_bytes") != CAMD_OK) return 0;
    return (size_t)g.gh * (4 + 8) + 64;
}

int camd_depth_to_point_cloud(const double* depth, int w, int h, const double Kinv[9], double rate, double* points,
                              double* uv, size_t capacity, unsigned long long* count, void* workspace, void* stream)
{
    PcGrid g;
    int rc = make_grid(&g, w, h, rate, "camd_depth_to_point_cloud");
    if (rc != CAMD_OK) return rc;
    if (!depth || !Kinv || !points || !count || !workspace) { set_error("camd_depth_to_point_cloud: NULL argument"); return CAMD_ERR_BAD_ARG; }
    rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* rowoff = reinterpret_cast<unsigned long long*>(workspace);
    uint32_t* rowcount = reinterpret_cast<uint32_t*>(rowoff + g.gh);
    hipLaunchKernelGGL(k_pc_count, dim3(g.gh), dim3(256), 0, st, depth, g, rowcount);
    hipLaunchKernelGGL(k_pc_scan, dim3(1), dim3(256), 0, st, rowcount, g.gh, rowoff, count);
    hipLaunchKernelGGL(k_pc_emit, dim3(g.gh), dim3(256), 0, st, depth, g, rowoff, Kinv[0], Kinv[1], Kinv[2], Kinv[3],
                       Kinv[4], Kinv[5], Kinv[6], Kinv[7], Kinv[8], points, uv, capacity);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_apply_T_to_point_cloud(const double* points, size_t n, const double T[16], double* out, void* stream)
{
    if (!T || (n && (!points || !out))) { set_error("camd_apply_T_to_point_cloud: NULL argument"); return CAMD_ERR_BAD_ARG; }
    if (n == 0) return CAMD_OK;
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    Mat34 M;
    for (int i = 0; i < 12; i++) M.m[i] = T[i];
    hipLaunchKernelGGL(k_apply_T, dim3(div_up((long long)n, 256)), dim3(256), 0, (hipStream_t)stream, points, n, M, out);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_point_cloud_to_depth(const double* points, size_t n, int point_stride, const double K[9], int w, int h,
                              double bg_value, double* depth, unsigned long long* keys_ws, void* stream)
{
    if (!K || !depth || !keys_ws || w <= 0 || h <= 0 || point_stride < 3 || (n && !points)) {
        set_error("camd_point_cloud_to_depth: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)w * h;
    Mat33 Km;
    for (int i = 0; i < 9; i++) Km.m[i] = K[i];
    hipLaunchKernelGGL(k_fill_u64, dim3(div_up((long long)npix, 256)), dim3(256), 0, st, keys_ws, npix, ZKEY_EMPTY);
    if (n) hipLaunchKernelGGL(k_pc_scatter, dim3(div_up((long long)n, 256)), dim3(256), 0, st, points, n, point_stride, Km, w, h, keys_ws);
    hipLaunchKernelGGL(k_pc_resolve, dim3(div_up((long long)npix, 256)), dim3(256), 0, st, keys_ws, npix, bg_value, depth);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_project_depth(const double* depth2, int w2, int h2, const double K2inv[9], const double T_2in1[16],
                       const double K1[9], double rate, int w1, int h1, double* depth1, unsigned long long* keys_ws,
                       void* stream)
{
    PcGrid g;
    int rc = make_grid(&g, w2, h2, rate, "camd_project_depth");
    if (rc != CAMD_OK) return rc;
    if (!depth2 || !K2inv || !T_2in1 || !K1 || !depth1 || !keys_ws || w1 <= 0 || h1 <= 0) {
        set_error("camd_project_depth: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)w1 * h1;
    Mat33 Ki, Km;
    Mat34 M;
    for (int i = 0; i < 9; i++) { Ki.m[i] = K2inv[i]; Km.m[i] = K1[i]; }
    for (int i = 0; i < 12; i++) M.m[i] = T_2in1[i];
    hipLaunchKernelGGL(k_fill_u64, dim3(div_up((long long)npix, 256)), dim3(256), 0, st, keys_ws, npix, ZKEY_EMPTY);
    hipLaunchKernelGGL(k_project_depth, dim3(div_up(g.gw, 256), g.gh), dim3(256), 0, st, depth2, g, Ki, M, Km, w1, h1, keys_ws);
    hipLaunchKernelGGL(k_pc_resolve, dim3(div_up((long long)npix, 256)), dim3(256), 0, st, keys_ws, npix, 0.0, depth1);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

}  // extern "C"

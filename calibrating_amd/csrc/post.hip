// post.hip -- the two filters cv2.StereoSGBM.compute applies after the dynamic programme:
//   medianBlur(disp, disp, 3)  (unconditional; int16, replicate border)
//   filterSpeckles(disp, (minD-1)*16, speckleWindowSize, 16*speckleRange)  iff speckleWindowSize > 0
// Reference call site: /root/reference/calibrating/stereo_matching.py:63 (inside .compute).
//
// Speckle filter = connected components of the graph whose edges join 4-neighbours a, b with
// a != newVal, b != newVal and |a - b| <= maxDiff; components of size <= maxSpeckleSize are
// overwritten with newVal.  The relation is symmetric, so the components do not depend on OpenCV's
// scan order and a lock-free union-find over horizontal runs gives the identical result.
#include "common.hpp"

namespace camd {

// exchange on two pixels at once: the signed 16-bit halves of a and b sorted independently
__device__ __forceinline__ void sort2(uint32_t& a, uint32_t& b)
{
    const uint32_t t = pk_min_i16(a, b);
    b = pk_max_i16(a, b);
    a = t;
}

// A thread owns the pixel pair (2i, 2i+1) of a row: its own pair arrives as one dword (global loads take any byte
// address on gfx950), the neighbour to the left and to the right as shorts, and the 19-exchange median network runs on
// packed int16 pairs (v_pk_min_i16 / v_pk_max_i16): half the loads and half the exchanges per pixel.
__global__ __launch_bounds__(256) void k_median3(const int16_t* __restrict__ src, size_t sp, size_t ss,
                                                 int16_t* __restrict__ dst, size_t dp, size_t ds, int w, int h)
{
    const int x = 2 * (blockIdx.x * 256 + threadIdx.x);
    const int y = blockIdx.y;
    if (x >= w) return;
    const bool two = x + 1 < w;  // (the last pair of an odd-width row holds one pixel)
    const int16_t* s = src + (size_t)blockIdx.z * ss;
    const int xl = x > 0 ? x - 1 : 0, xr = min(x + 2, w - 1);
    const int16_t* rows[3] = {s + (size_t)(y > 0 ? y - 1 : y) * sp, s + (size_t)y * sp,
                              s + (size_t)(y < h - 1 ? y + 1 : y) * sp};
    uint32_t p[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        uint32_t own;
        if (two) __builtin_memcpy(&own, rows[r] + x, 4);
        else own = dup16((uint16_t)rows[r][x]);  // replicate border: the missing right neighbour is the pixel itself
        const uint32_t l = (uint16_t)rows[r][xl], rr = (uint16_t)rows[r][xr];
        p[3 * r + 0] = l | (own << 16);           // left neighbours of (x, x+1):   (x-1, x)
        p[3 * r + 1] = own;                       //                                 (x,   x+1)
        p[3 * r + 2] = (own >> 16) | (rr << 16);  // right neighbours:              (x+1, x+2)
    }
    // median of 9 by the classic 19-exchange network
    sort2(p[1], p[2]); sort2(p[4], p[5]); sort2(p[7], p[8]); sort2(p[0], p[1]);
    sort2(p[3], p[4]); sort2(p[6], p[7]); sort2(p[1], p[2]); sort2(p[4], p[5]);
    sort2(p[7], p[8]); sort2(p[0], p[3]); sort2(p[5], p[8]); sort2(p[4], p[7]);
    sort2(p[3], p[6]); sort2(p[1], p[4]); sort2(p[2], p[5]); sort2(p[4], p[7]);
    sort2(p[4], p[2]); sort2(p[6], p[4]); sort2(p[4], p[2]);
    int16_t* o = dst + (size_t)blockIdx.z * ds + (size_t)y * dp + x;
    if (two) __builtin_memcpy(o, &p[4], 4);
    else *o = (int16_t)(p[4] & 0xffffu);
}

int launch_median3(const int16_t* src, size_t sp, size_t ss, int16_t* dst, size_t dp, size_t ds, int w, int h,
                   int batch, hipStream_t st)
{
    hipLaunchKernelGGL(k_median3, dim3(div_up(div_up(w, 2), 256), h, batch), dim3(256), 0, st, src, sp, ss, dst, dp, ds,
                       w, h);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

// ---- union-find connected components --------------------------------------------------------------
__device__ __forceinline__ int uf_find(int* parent, int a)
{
    int p = parent[a];
    while (p != a) {
        int gp = parent[p];
        if (gp != p) parent[a] = gp;  // path halving (benign race: only ever points closer to the root)
        a = p;
        p = gp;
    }
    return a;
}

__device__ void uf_union(int* parent, int a, int b)
{
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }  // a > b: hook the larger root under the smaller
        int old = atomicCAS(&parent[a], a, b);
        if (old == a) return;
        a = old;
    }
}

// All kernels: blockIdx.z = image of the batch; parent / count hold 2*n ints per image (labels are local); a block is
// 256 consecutive pixels of one row, so a wave is 64 consecutive pixels.
//
// Horizontal structure comes for free: within a wave the pixels of a horizontal run (each connected to its left
// neighbour) are found from one ballot, so
//   k_cc_rows   points every pixel at the first pixel of its run-in-the-wave (no atomics at all),
//   k_cc_merge  unions only (i) a run that continues across a wave boundary with the wave before and (ii) vertical
//               neighbours -- and skips a vertical edge whenever the edge one pixel to the left together with the two
//               horizontal edges already implies it (in smooth regions that leaves one union per wave and row),
//   k_cc_count  adds the LENGTH of each run to its root with one atomic per run instead of one per pixel (a large
//               component no longer serialises on a single counter),
//   k_cc_apply  erases the components that are small enough.
struct CcPix {
    int v;          // pixel value
    bool valid;     // != newVal
    bool cl;        // connected to the left neighbour (same row)
};

__device__ __forceinline__ CcPix cc_load(const int16_t* __restrict__ row, int x, int w, int new_val, int max_diff)
{
    CcPix p;
    p.v = x < w ? row[x] : new_val;
    p.valid = p.v != new_val;
    const int l = dpp_perm<DPP_WAVE_SHR1>((uint32_t)p.v);  // lane 0 is handled by the caller
    p.cl = p.valid && l != new_val && abs(p.v - l) <= max_diff;
    return p;
}

// lane of the first pixel of this lane's run inside the wave, and whether the run reaches back past lane 0
__device__ __forceinline__ int cc_run_start(unsigned long long clmask, int lane)
{
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    const unsigned long long breaks = ~clmask & upto;  // lanes <= lane that do NOT connect to their left
    return breaks ? 63 - __clzll(breaks) : 0;
}

__global__ __launch_bounds__(256) void k_cc_rows(const int16_t* __restrict__ img, size_t pitch, size_t stride,
                                                 int* parent, int w, int h, int new_val, int max_diff)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lane = threadIdx.x & 63;
    img += (size_t)blockIdx.z * stride;
    parent += (size_t)blockIdx.z * 2 * w * h;
    CcPix p = cc_load(img + (size_t)y * pitch, x, w, new_val, max_diff);
    if (lane == 0) p.cl = false;  // the link to the previous wave is an explicit union in k_cc_merge
    const unsigned long long m = __ballot(p.cl);
    if (x < w) {
        const int i = y * w + x, start = cc_run_start(m, lane);
        parent[i] = i - (lane - start);
        // counts live at roots, and a root is always the first pixel of a run-in-the-wave (unions hook the larger
        // root under the smaller, so roots stay among the initial ones): only those are zeroed
        if (start == lane) parent[w * h + i] = 0;
    }
}

__global__ __launch_bounds__(256) void k_cc_merge(const int16_t* __restrict__ img, size_t pitch, size_t stride,
                                                  int* parent, int w, int h, int new_val, int max_diff)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lane = threadIdx.x & 63;
    img += (size_t)blockIdx.z * stride;
    parent += (size_t)blockIdx.z * 2 * w * h;
    const int16_t* row = img + (size_t)y * pitch;
    CcPix p = cc_load(row, x, w, new_val, max_diff);
    // the row below, same columns
    int d = new_val;
    if (y + 1 < h && x < w) d = row[pitch + x];
    const bool cd = p.valid && d != new_val && abs(p.v - d) <= max_diff;         // vertical edge (x, y)-(x, y+1)
    const int dl = dpp_perm<DPP_WAVE_SHR1>((uint32_t)d);
    const bool cl_down = d != new_val && dl != new_val && abs(d - dl) <= max_diff;  // (x-1, y+1)-(x, y+1)
    const bool cd_left = dpp_perm<DPP_WAVE_SHR1>((uint32_t)cd) != 0;              // vertical edge one pixel to the left
    if (x >= w || !p.valid) return;
    const int i = y * w + x;
    if (lane == 0) {
        // across the wave boundary: lane 0 has no DPP neighbour, so it looks at memory
        if (x > 0) {
            const int l = row[x - 1];
            if (l != new_val && abs(p.v - l) <= max_diff) uf_union(parent, i, i - 1);
        }
        if (cd) uf_union(parent, i, i + w);
    } else if (cd && !(p.cl && cd_left && cl_down)) {
        uf_union(parent, i, i + w);
    }
}

__global__ __launch_bounds__(256) void k_cc_count(const int16_t* __restrict__ img, size_t pitch, size_t stride,
                                                  int* parent, int w, int h, int new_val, int max_diff)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lane = threadIdx.x & 63;
    img += (size_t)blockIdx.z * stride;
    parent += (size_t)blockIdx.z * 2 * w * h;
    int* count = parent + w * h;
    CcPix p = cc_load(img + (size_t)y * pitch, x, w, new_val, max_diff);
    if (lane == 0) p.cl = false;
    const unsigned long long m = __ballot(p.cl);
    if (x >= w || !p.valid || p.cl) return;  // only the first pixel of a run-in-the-wave counts, for the whole run
    const unsigned long long after = lane == 63 ? 0ull : (m >> (lane + 1));
    const int len = 1 + (~after ? __ffsll((long long)~after) - 1 : 64);  // consecutive connected lanes behind it
    const int i = y * w + x;
    const int r = uf_find(parent, i);
    parent[i] = r;
    atomicAdd(&count[r], len);
}

// one walk to the root per run-in-the-wave: its first pixel looks the component's size up, the others take its verdict
__global__ __launch_bounds__(256) void k_cc_apply(int16_t* img, size_t pitch, size_t stride,
                                                  const int* __restrict__ parent, int w, int h, int new_val,
                                                  int max_size, int max_diff)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lane = threadIdx.x & 63;
    img += (size_t)blockIdx.z * stride;
    parent += (size_t)blockIdx.z * 2 * w * h;
    const int* count = parent + w * h;
    CcPix p = cc_load(img + (size_t)y * pitch, x, w, new_val, max_diff);
    if (lane == 0) p.cl = false;
    const unsigned long long m = __ballot(p.cl);
    const int start = cc_run_start(m, lane);
    int kill = 0;
    if (x < w && p.valid && start == lane) {
        // first pixel of its run -> (compressed by k_cc_count) root; walk read-only
        int r = y * w + x;
        for (int q = parent[r]; q != r; q = parent[r]) r = q;
        kill = count[r] <= max_size;
    }
    kill = __shfl(kill, start);
    if (x < w && p.valid && kill) img[(size_t)y * pitch + x] = (int16_t)new_val;
}

size_t speckle_ws_bytes(int w, int h, int batch)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    return (size_t)batch * w * h * 2 * sizeof(int);  // parent + count per image: the whole batch in one launch
}

int launch_speckle(int16_t* img, size_t pitch_e, size_t stride_e, int w, int h, int new_val, int max_size,
                   int max_diff, void* ws, int batch, hipStream_t st)
{
    if (!ws) { set_error("speckle workspace is NULL"); return CAMD_ERR_BAD_ARG; }
    int n = w * h;
    int* parent = reinterpret_cast<int*>(ws);
    dim3 grid(div_up(w, 256), h, batch);
    (void)n;
    hipLaunchKernelGGL(k_cc_rows, grid, dim3(256), 0, st, img, pitch_e, stride_e, parent, w, h, new_val, max_diff);
    hipLaunchKernelGGL(k_cc_merge, grid, dim3(256), 0, st, img, pitch_e, stride_e, parent, w, h, new_val, max_diff);
    hipLaunchKernelGGL(k_cc_count, grid, dim3(256), 0, st, img, pitch_e, stride_e, parent, w, h, new_val, max_diff);
    hipLaunchKernelGGL(k_cc_apply, grid, dim3(256), 0, st, img, pitch_e, stride_e, parent, w, h, new_val, max_size,
                       max_diff);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

}  // namespace camd

using namespace camd;

extern "C" {

int camd_median3_s16(const int16_t* src, int16_t* dst, int w, int h, int batch, void* stream)
{
    if (!src || !dst || w <= 0 || h <= 0 || batch <= 0 || src == dst) {
        set_error("camd_median3_s16: bad arguments (dst must differ from src)");
        return CAMD_ERR_BAD_ARG;
    }
    return launch_median3(src, w, (size_t)w * h, dst, w, (size_t)w * h, w, h, batch, (hipStream_t)stream);
}

size_t camd_speckle_workspace_bytes(int w, int h, int batch) { return speckle_ws_bytes(w, h, batch); }

int camd_filter_speckles_s16(int16_t* img, int w, int h, int new_val, int max_speckle_size, int max_diff,
                             void* labels_ws, int batch, void* stream)
{
    if (!img || w <= 0 || h <= 0 || batch <= 0) {
        set_error("camd_filter_speckles_s16: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    return launch_speckle(img, w, (size_t)w * h, w, h, new_val, max_speckle_size, max_diff, labels_ws, batch,
                          (hipStream_t)stream);
}

}  // extern "C"

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
// packed int16 pairs (v_pk_min_i16 / v_pk_max_i16): half the loads and half the exchanges per pixel.  A block walks
// POST_ROWS rows and keeps the three rows of the window in registers (one new row per output row; one row per block,
// the round-3 form, was bound by the rate at which 250 K tiny workgroups start).
constexpr int POST_ROWS = 8;

__global__ __launch_bounds__(256) void k_median3(const int16_t* __restrict__ src, size_t sp, size_t ss,
                                                 int16_t* __restrict__ dst, size_t dp, size_t ds, int w, int h)
{
    const int x = 2 * (blockIdx.x * 256 + threadIdx.x);
    const int y0 = blockIdx.y * POST_ROWS, y1 = min(y0 + POST_ROWS, h);
    if (x >= w) return;
    const bool two = x + 1 < w;  // (the last pair of an odd-width row holds one pixel)
    const int16_t* s = src + (size_t)blockIdx.z * ss;
    const int xl = x > 0 ? x - 1 : 0, xr = min(x + 2, w - 1);
    uint32_t win[3][3];  // [row of the window][left neighbours | own pair | right neighbours]
    auto load_row = [&](int y, uint32_t (&r)[3]) {
        const int16_t* row = s + (size_t)min(max(y, 0), h - 1) * sp;  // replicate border
        uint32_t own;
        if (two) __builtin_memcpy(&own, row + x, 4);
        else own = dup16((uint16_t)row[x]);  // replicate border: the missing right neighbour is the pixel itself
        const uint32_t l = (uint16_t)row[xl], rr = (uint16_t)row[xr];
        r[0] = l | (own << 16);           // left neighbours of (x, x+1):   (x-1, x)
        r[1] = own;                       //                                 (x,   x+1)
        r[2] = (own >> 16) | (rr << 16);  // right neighbours:              (x+1, x+2)
    };
    load_row(y0 - 1, win[0]);
    load_row(y0, win[1]);
#pragma unroll
    for (int k = 0; k < POST_ROWS; k++) {
        const int y = y0 + k;
        if (y >= y1) break;
        load_row(y + 1, win[(k + 2) % 3]);
        uint32_t p[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) p[3 * r + c] = win[(k + r) % 3][c];
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
}

int launch_median3(const int16_t* src, size_t sp, size_t ss, int16_t* dst, size_t dp, size_t ds, int w, int h,
                   int batch, hipStream_t st)
{
    hipLaunchKernelGGL(k_median3, dim3(div_up(div_up(w, 2), 256), div_up(h, POST_ROWS), batch), dim3(256), 0, st, src, sp,
                       ss, dst, dp, ds, w, h);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

// ---- union-find connected components --------------------------------------------------------------
// Layout of the workspace: per image 2 * w * h ints = parent[w * h] then count[w * h]; node ids are pixel indices
// local to the image.  INVARIANT between calls: every parent entry is -1 ("root of its own tree").  The kernels below
// only ever write the entries of nodes -- the first pixel of a horizontal run inside a 64-pixel segment ("run start")
// and a segment's last pixel -- and k_cc_apply puts every one of them back to -1, so nothing has to be initialised
// per call (the owner of the workspace clears it once; see launch_speckle's `clean`).
__device__ __forceinline__ int uf_find(int* parent, int a)
{
    while (true) {
        const int p = parent[a];
        if (p < 0) return a;
        const int gp = parent[p];
        if (gp >= 0) parent[a] = gp;  // path halving (benign race: only ever points closer to the root)
        a = p;
    }
}

__device__ __forceinline__ int uf_find_ro(const int* parent, int a)
{
    for (int p = parent[a]; p >= 0; p = parent[a]) a = p;
    return a;
}

__device__ void uf_union(int* parent, int a, int b)
{
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }  // a > b: hook the larger root under the smaller
        const int old = atomicCAS(&parent[a], -1, b);
        if (old < 0) return;
        a = old;  // someone hooked a first (or this lane's view of it was stale): carry on from its parent
    }
}

// Geometry of all four kernels: a wave owns a SEGMENT of 64 columns and walks a STRIP of CC_ROWS rows top to bottom
// (a block = 4 neighbouring segments; blockIdx.y = strip, blockIdx.z = image).  One row per wave and launch -- the
// round-3 form -- was bound by the rate at which waves start, not by memory: 553 K four-wave blocks of ~20
// instructions per kernel.
//
// Horizontal structure comes for free: inside a segment the pixels of a horizontal run (each connected to its left
// neighbour) are found from one ballot.  Vertical structure inside a strip is carried in registers: a run takes the
// label (= node id of a run start further up) of the first run above it that it touches, a run that touches nothing
// above is "born" as the root of a new tree, and a global union is needed only where a run touches runs with
// DIFFERENT labels.  In smooth regions that leaves no atomic at all inside a strip.
//   k_cc_label    strip-local labelling; parent[run start] = label (plain store; born runs stay -1), count = 0 at
//                 born runs (every root is one: all other nodes get a parent here), parent[segment's last pixel] =
//                 its run start so that the neighbouring segment can name it
//   k_cc_hborders / k_cc_vborders  unions across the strips' horizontal borders and the segments' vertical borders,
//                 skipping every edge that the edge before it together with the two links along the border implies
//   k_cc_count    per run start: root (read-only walk), parent[start] = root, run length added to count[root] --
//                 accumulated down the strip while the root stays the same, and no longer added once the count
//                 is past max_size (only `count <= max_size` is ever asked): big components cost a few atomics
//   k_cc_apply    erases the runs whose root's count is small enough and puts parent back to -1
constexpr int CC_ROWS = 16;

// lane of the first pixel of this lane's run inside the segment
__device__ __forceinline__ int cc_run_start(unsigned long long clmask, int lane)
{
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    const unsigned long long breaks = ~clmask & upto;  // lanes <= lane that do NOT connect to their left
    return breaks ? 63 - __clzll(breaks) : 0;
}
// lane of its last pixel
__device__ __forceinline__ int cc_run_end(unsigned long long clmask, int lane)
{
    const unsigned long long above = lane == 63 ? 0ull : (~clmask & (~0ull << (lane + 1)));
    return above ? __ffsll((long long)above) - 2 : 63;
}
__device__ __forceinline__ bool cc_close(int a, int b, int new_val, int max_diff)
{
    return a != new_val && b != new_val && abs(a - b) <= max_diff;
}
// connected-to-the-left flag of a row held one pixel per lane (lane 0: never -- segments are joined by k_cc_borders)
__device__ __forceinline__ bool cc_left(int v, int lane, int new_val, int max_diff)
{
    const int l = (int)dpp_perm<DPP_WAVE_SHR1>((uint32_t)v);
    return lane != 0 && cc_close(v, l, new_val, max_diff);
}

// Every strip kernel starts the same way: all CC_ROWS rows of the segment are requested at once (a wave walking its
// rows one load at a time spends the pass waiting: 16 dependent round trips), rows below the image read as new_val.
#define CC_STRIP_PROLOGUE()                                                                          \
    const int lane = threadIdx.x & 63;                                                               \
    const int x = blockIdx.x * 256 + threadIdx.x;                                                    \
    const int y0 = blockIdx.y * CC_ROWS;                                                             \
    img += (size_t)blockIdx.z * stride;                                                              \
    parent += (size_t)blockIdx.z * 2 * w * h;                                                        \
    if (blockIdx.x * 256 + (threadIdx.x & ~63) >= w) return; /* whole segment outside the image */   \
    int vrow[CC_ROWS];                                                                               \
    _Pragma("unroll") for (int k = 0; k < CC_ROWS; k++)                                              \
        vrow[k] = (x < w && y0 + k < h) ? img[(size_t)(y0 + k) * pitch + x] : new_val;

// What a strip kernel knows about the run of its lane in the current row
struct CcRun {
    bool valid;                // pixel != newVal
    int start, end;            // lanes of the run's first / last pixel
    unsigned long long m;      // connected-to-the-left mask of the row
    unsigned long long cand;   // pixels of this run that touch the row above (inside the strip)
    int first;                 // lane of the first of them (own lane if none)
    bool stored;               // k_cc_label stores a parent entry for this run's start: its run is born (then the entry is
                               // the union-find's), or another segment / strip may have to name it
};
__device__ __forceinline__ CcRun cc_run(int v, int v_up, int lane, bool last_row, int new_val, int max_diff)
{
    CcRun r;
    r.valid = v != new_val;
    r.m = __ballot(cc_left(v, lane, new_val, max_diff));
    const unsigned long long mu = __ballot(cc_close(v, v_up, new_val, max_diff));
    r.start = cc_run_start(r.m, lane);
    r.end = cc_run_end(r.m, lane);
    const unsigned long long runmask = (r.end == 63 ? ~0ull : ((2ull << r.end) - 1)) & (~0ull << r.start);
    r.cand = mu & runmask;
    r.first = r.cand ? __ffsll((long long)r.cand) - 1 : lane;
    r.stored = !r.cand || r.start == 0 || r.end == 63 || last_row;
    return r;
}

__global__ __launch_bounds__(256) void k_cc_label(const int16_t* __restrict__ img, size_t pitch, size_t stride,
                                                  int* parent, int w, int h, int new_val, int max_diff)
{
    CC_STRIP_PROLOGUE();
    int* count = parent + w * h;
    int v_up = new_val, lab_up = -1;
    unsigned long long m_up = 0;
#pragma unroll
    for (int k = 0; k < CC_ROWS; k++) {
        const int y = y0 + k, v = vrow[k];
        if (y >= h) break;
        const CcRun r = cc_run(v, v_up, lane, k == CC_ROWS - 1, new_val, max_diff);
        const int i = y * w + x, self = i - (lane - r.start);
        int label = __shfl(lab_up, r.first);
        if (!r.cand) label = self;
        if (r.valid) {
            if (lane == r.start) {
                if (!r.cand) count[i] = 0;
                else if (r.stored) parent[i] = label;
            } else if (lane == 63) {
                parent[i] = self;
            }
            const bool cu = (r.cand >> lane) & 1;
            if (cu && lab_up != label) {
                // this run touches a second tree; one lane per overlap of the two runs reports it
                const bool implied = lane > 0 && ((r.m >> lane) & 1) && ((r.cand >> (lane - 1)) & 1) && ((m_up >> lane) & 1);
                if (!implied) uf_union(parent, lab_up, label);
            }
        }
        v_up = v;
        lab_up = r.valid ? label : -1;
        m_up = r.m;
    }
}

// horizontal border above strip (blockIdx.y + 1): 4 segments per block
__global__ __launch_bounds__(256) void k_cc_hborders(const int16_t* __restrict__ img, size_t pitch, size_t stride,
                                                     int* parent, int w, int h, int new_val, int max_diff)
{
    const int lane = threadIdx.x & 63;
    img += (size_t)blockIdx.z * stride;
    parent += (size_t)blockIdx.z * 2 * w * h;
    const int x = blockIdx.x * 256 + threadIdx.x, yd = (blockIdx.y + 1) * CC_ROWS, yu = yd - 1;
    if (blockIdx.x * 256 + (threadIdx.x & ~63) >= w) return;
    const int u = x < w ? img[(size_t)yu * pitch + x] : new_val;
    const int d = x < w ? img[(size_t)yd * pitch + x] : new_val;
    const bool clu = cc_left(u, lane, new_val, max_diff), cld = cc_left(d, lane, new_val, max_diff);
    const unsigned long long mu = __ballot(clu), md = __ballot(cld);
    const bool cv = cc_close(u, d, new_val, max_diff);
    const bool cv_left = dpp_perm<DPP_WAVE_SHR1>((uint32_t)cv) != 0;
    if (cv && !(lane > 0 && clu && cld && cv_left))
        uf_union(parent, yu * w + x - (lane - cc_run_start(mu, lane)), yd * w + x - (lane - cc_run_start(md, lane)));
}

// vertical borders: a wave = 64 consecutive rows of the border left of segment sb >= 1 (lanes = rows); a = last pixel of
// the segment to the left (a node: k_cc_label pointed it at its run start), b = first pixel of this segment (always a
// run start)
__global__ __launch_bounds__(256) void k_cc_vborders(const int16_t* __restrict__ img, size_t pitch, size_t stride,
                                                     int* parent, int w, int h, int new_val, int max_diff, int nseg)
{
    const int lane = threadIdx.x & 63;
    img += (size_t)blockIdx.z * stride;
    parent += (size_t)blockIdx.z * 2 * w * h;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nrow64 = (h + 63) >> 6;
    const int sb = 1 + wave / nrow64;
    const int y = (wave % nrow64) * 64 + lane;
    if (sb >= nseg) return;
    const int xb = sb * 64;
    const bool in = y < h;
    const int a = in ? img[(size_t)y * pitch + xb - 1] : new_val;
    const int b = in ? img[(size_t)y * pitch + xb] : new_val;
    const bool ch = cc_close(a, b, new_val, max_diff);
    const int a_up = (int)dpp_perm<DPP_WAVE_SHR1>((uint32_t)a), b_up = (int)dpp_perm<DPP_WAVE_SHR1>((uint32_t)b);
    const bool ch_up = dpp_perm<DPP_WAVE_SHR1>((uint32_t)ch) != 0;
    const bool implied = lane > 0 && ch_up && cc_close(a, a_up, new_val, max_diff) && cc_close(b, b_up, new_val, max_diff);
    if (ch && !implied) uf_union(parent, y * w + xb - 1, y * w + xb);
}

__global__ __launch_bounds__(256) void k_cc_count(const int16_t* __restrict__ img, size_t pitch, size_t stride,
                                                  int* parent, int w, int h, int new_val, int max_size, int max_diff)
{
    CC_STRIP_PROLOGUE();
    int* count = parent + w * h;
    int acc_root = -1, acc_n = 0;
    auto flush = [&]() {
        if (acc_n > 0 && __hip_atomic_load(&count[acc_root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= max_size)
            atomicAdd(&count[acc_root], acc_n);
    };
    int v_up = new_val, root_up = -1;
#pragma unroll
    for (int k = 0; k < CC_ROWS; k++) {
        const int y = y0 + k, v = vrow[k];
        if (y >= h) break;
        const CcRun r = cc_run(v, v_up, lane, k == CC_ROWS - 1, new_val, max_diff);
        const int i = y * w + x;
        // a run that touches the row above has that run's root (all unions are done); only a born run walks
        int root = __shfl(root_up, r.first);
        if (r.valid && !r.cand && lane == r.start) {
            root = uf_find_ro(parent, i);
            if (root != i) parent[i] = root;  // k_cc_apply reads the root in one step
        }
        const int born_root = __shfl(root, r.start);
        if (!r.cand) root = born_root;
        if (r.valid && lane == r.start) {  // one lane per run
            const int len = r.end - lane + 1;
            if (root == acc_root) {
                acc_n += len;
            } else {
                flush();
                acc_root = root;
                acc_n = len;
            }
        }
        v_up = v;
        root_up = r.valid ? root : -1;
    }
    flush();
}

__global__ __launch_bounds__(256) void k_cc_apply(int16_t* img, size_t pitch, size_t stride, int* parent, int w, int h,
                                                  int new_val, int max_size, int max_diff)
{
    CC_STRIP_PROLOGUE();
    const int* count = parent + w * h;
    int v_up = new_val, kill_up = 0;
#pragma unroll
    for (int k = 0; k < CC_ROWS; k++) {
        const int y = y0 + k, v = vrow[k];
        if (y >= h) break;
        const CcRun r = cc_run(v, v_up, lane, k == CC_ROWS - 1, new_val, max_diff);
        const int i = y * w + x;
        int kill = __shfl(kill_up, r.first);  // the verdict of the run above, if there is one
        if (r.valid && lane == r.start) {
            if (!r.cand) {
                const int p = parent[i];  // k_cc_count left the root here (-1: this run start is the root itself)
                kill = count[p < 0 ? i : p] <= max_size;
                if (p >= 0) parent[i] = -1;
            } else if (r.stored) {
                parent[i] = -1;
            }
        } else if (r.valid && lane == 63) {
            parent[i] = -1;
        }
        const int born_kill = __shfl(kill, r.start);
        if (!r.cand) kill = born_kill;
        if (r.valid && kill) img[(size_t)y * pitch + x] = (int16_t)new_val;
        v_up = v;
        kill_up = r.valid ? kill : 0;
    }
}

size_t speckle_ws_bytes(int w, int h, int batch)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    return (size_t)batch * w * h * 2 * sizeof(int);  // parent + count per image: the whole batch in one launch
}

// `clean` (may be NULL): in -- the WHOLE workspace (`ws_bytes`: it may be sized for more images than this call's
// `batch`) already holds -1 in every parent entry (it was cleared once and every call since completed); out -- true once
// all kernels are queued.  When it is not known to be clean all of it is cleared first (memset to 0xFF: parent = -1;
// count is zeroed where it is used).
int launch_speckle(int16_t* img, size_t pitch_e, size_t stride_e, int w, int h, int new_val, int max_size,
                   int max_diff, void* ws, size_t ws_bytes, int batch, hipStream_t st, bool* clean)
{
    if (!ws) { set_error("speckle workspace is NULL"); return CAMD_ERR_BAD_ARG; }
    if (ws_bytes < speckle_ws_bytes(w, h, batch)) { set_error("speckle workspace too small for %d images", batch); return CAMD_ERR_BAD_ARG; }
    int* parent = reinterpret_cast<int*>(ws);
    if (!clean || !*clean) CAMD_HIP(hipMemsetAsync(ws, 0xFF, ws_bytes, st));
    if (clean) *clean = false;
    const int nseg = div_up(w, 64), nstrips = div_up(h, CC_ROWS), hblocks = div_up(w, 256);
    const int vwaves = (nseg - 1) * div_up(h, 64);
    dim3 grid(hblocks, nstrips, batch);
    hipLaunchKernelGGL(k_cc_label, grid, dim3(256), 0, st, img, pitch_e, stride_e, parent, w, h, new_val, max_diff);
    if (nstrips > 1)
        hipLaunchKernelGGL(k_cc_hborders, dim3(hblocks, nstrips - 1, batch), dim3(256), 0, st, img, pitch_e, stride_e,
                           parent, w, h, new_val, max_diff);
    if (vwaves > 0)
        hipLaunchKernelGGL(k_cc_vborders, dim3(div_up(vwaves, 4), 1, batch), dim3(256), 0, st, img, pitch_e, stride_e,
                           parent, w, h, new_val, max_diff, nseg);
    hipLaunchKernelGGL(k_cc_count, grid, dim3(256), 0, st, img, pitch_e, stride_e, parent, w, h, new_val, max_size, max_diff);
    hipLaunchKernelGGL(k_cc_apply, grid, dim3(256), 0, st, img, pitch_e, stride_e, parent, w, h, new_val, max_size,
                       max_diff);
    CAMD_LAUNCH_CHECK();
    if (clean) *clean = true;
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
    return launch_speckle(img, w, (size_t)w * h, w, h, new_val, max_speckle_size, max_diff, labels_ws,
                          speckle_ws_bytes(w, h, batch), batch, (hipStream_t)stream, nullptr);  // a caller's scratch: nothing is known about its contents
}

}  // extern "C"

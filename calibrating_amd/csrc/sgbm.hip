// sgbm.hip -- Semi-Global Block Matching for gfx950 (MI355X), bit-exact with cv2.StereoSGBM
// (modes MODE_SGBM, MODE_HH, MODE_HH4 and MODE_SGBM_3WAY) at equal parameters.
//
// Replaces the cv2.StereoSGBM_create(...).compute(left, right) call of the reference
// (/root/reference/calibrating/stereo_matching.py:48-58,63).  Not a port of OpenCV's row-incremental
// CPU loop: the algorithm is re-stated in a data-parallel form (SURVEY.md Appendix A / DESIGN.md):
//
//   k_cost         (sgbm_cost.hpp) calcPixelCostBT + blockSize x blockSize box sum + P2 -> C[y][x][d] in ONE pass:
//                  lanes = columns, the workgroup walks rows; horizontal sum by wave-wide DPP shifts, vertical sum as a
//                  register ring.  The default for blockSize <= 11.
//   k_hsum, k_vsum the two-pass form of the same (round 1): BT + horizontal box sum into an intermediate volume, then
//                  the vertical box sum + P2.  Kept for blockSize 13 / 15 and as an A/B reference (CAMD_COST_SPLIT).
//   k_band         (sgbm_band.hpp) fused aggregation: four directions per pass + WTA in the last
//   k_scan         one aggregation direction as independent line scans from a zero border state;
//                  a line is owned by a 2..16-lane group (2*NR disparities per lane, packed u16x2),
//                  neighbours d-1 / d+1 by DPP row shifts, min over d by a DPP butterfly
//   k_wta          winner-take-all, uniqueness, sub-pixel parabola, right-view map via LDS
//                  atomicMin on (cost << 16 | 0xFFFF - d), left-right check; one workgroup per row
//   k_median3 / speckle (post.hip)
//
// All volumes are [pair][y][x][Dp] int16 with Dp = LANES*2*NR >= D (NR packed u16x2 registers per lane); entries
// d >= D carry MAX_COST.
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <new>

namespace camd {

static constexpr uint32_t SENT_PK = 0x7fff7fffu;  // MAX_COST in both halves
static constexpr int MAX_COST = 32767;
static constexpr int CAMD_MULTI_MAX_BATCH = 8;  // concurrent-direction path: npaths volumes for up to 8 pairs per call

struct Geom {
    int W, H, cn;          // image
    int minD, D, Dp;       // disparity range, padded
    int minX1, W1;         // cost coordinates: x_img = x + minX1
    int SW2;               // box radius
    int P1, P2, uniq, d12; // normalised parameters
    int ftzero;
    int lanes, nr;         // line-group shape: a lane holds nr packed registers = 2*nr disparities, Dp = lanes*2*nr
    int mode, npaths;
    uint32_t uniq_magic;   // floor(2^32 / (100 - uniq)) + 1, or 0 when 100 - uniq == 1 (band WTA)
    int speckleWindowSize, speckleRange;
};

// ------------------------------------------------------------------------------------------------
// k_hsum: Hs[y][x][d] = sum_{dx=-SW2..SW2} pix(y, clamp(x+dx, 0, W1-1), d)   (u16, wraps like
// OpenCV's CostType), pix = sum over channels of min(c0, c1) [gradient] + min(c0, c1) >> 2 [raw].
// ------------------------------------------------------------------------------------------------
static constexpr int HSUM_SEG = 128;   // cost columns per workgroup
static constexpr int HSUM_RING = 16;   // ring slots (>= 2*SW2+1), per lane, in LDS

// One workgroup = one row y, cost columns [xs, xe), ALL disparities: wave w owns d in [64w, 64w+64).
// The right-image operands of the row segment are staged once in LDS (entry = one right pixel,
// CN*3 dwords padded to a multiple of 16 bytes); lane j of wave w reads entry (c - clo) + last - d for
// cost column c, so every step costs three ds_read_b128 (RGB) instead of VALU shifts.  The left-image
// operands are wave-uniform scalar loads.
// KT = the box width 2*SW2+1 when it is one of the instantiated sizes: the per-lane ring of the last KT column
// costs then lives in registers (the step loop is unrolled by 2*KT so every slot index is static);
// KT = 0: any width, ring in LDS.
template <int CN, int KT>
__global__ __launch_bounds__(512) void k_hsum(const uint8_t* __restrict__ left,
                                              const uint8_t* __restrict__ right, size_t pitch,
                                              size_t image_stride, uint16_t* __restrict__ Hs, Geom g,
                                              int ndblk, size_t vol_stride)
{
    constexpr int ES = CN == 1 ? 4 : 12;  // dwords per staged pixel
    extern __shared__ __attribute__((aligned(16))) uint32_t hs_lds[];
    const int lane = threadIdx.x & 63, dblk = threadIdx.x >> 6;
    const int y = blockIdx.y, pair = blockIdx.z;
    const int xs = blockIdx.x * HSUM_SEG, xe = min(xs + HSUM_SEG, g.W1);
    const int d = dblk * 64 + lane;
    const int K = 2 * g.SW2 + 1;
    const int clo = max(xs - g.SW2, 0), chi = min(xe - 1 + g.SW2, g.W1 - 1);
    const int last = ndblk * 64 - 1;
    const int ncols = (chi - clo) + ndblk * 64;          // right-image entries
    const int nleft = chi - clo + 1;                      // left-image entries (cost columns clo..chi)
    const int colbase = clo + g.minX1 - g.minD - last;    // right-image column of entry 0
    const int maxr = HSUM_SEG + 2 * g.SW2 + ndblk * 64 + 2, maxl = HSUM_SEG + 2 * g.SW2 + 2;
    uint32_t* stage = hs_lds;                              // [maxr][ES]  right operands (+1 halo entry each side)
    uint32_t* lstage = stage + (size_t)maxr * ES;          // [maxl][ES]  left operands  (+1 halo entry each side)
    uint32_t* ring = lstage + (size_t)maxl * ES + (size_t)dblk * K * 64;  // [ndblk][K][64]

    // ---- fused calcPixelCostBT preprocessing: planes p = (clipped x-Sobel | raw << 16) of the image
    // columns this segment touches, then per entry (p, min(p,(p+l)/2,(p+r)/2), max(...)).  Columns 0 and
    // W-1 of every plane hold ftzero; at the image edge the missing neighbour is p itself.
    // Phase 1 writes p into slot 0 of every entry (halo included), phase 2 reads the neighbours' slot 0.
    const uint32_t ftz2 = (uint32_t)g.ftzero | ((uint32_t)g.ftzero << 16);
    auto plane = [&](const uint8_t* img, int col, int c) -> uint32_t {
        if (col <= 0 || col >= g.W - 1) return ftz2;  // also covers columns outside the image (unused d)
        const uint8_t* r0 = img + (size_t)y * pitch + (size_t)col * CN + c;
        const uint8_t* rm = img + (size_t)(y > 0 ? y - 1 : y) * pitch + (size_t)col * CN + c;
        const uint8_t* rp = img + (size_t)(y < g.H - 1 ? y + 1 : y) * pitch + (size_t)col * CN + c;
        int gq = ((int)r0[CN] - (int)r0[-CN]) * 2 + ((int)rm[CN] - (int)rm[-CN]) + ((int)rp[CN] - (int)rp[-CN]);
        gq = min(max(gq, -g.ftzero), g.ftzero) + g.ftzero;
        return (uint32_t)gq | ((uint32_t)r0[0] << 16);
    };
    const uint8_t* imgR = right + (size_t)pair * image_stride;
    const uint8_t* imgL = left + (size_t)pair * image_stride;
    for (int e = threadIdx.x; e < ncols + 2; e += blockDim.x)
#pragma unroll
        for (int c = 0; c < CN; c++) stage[e * ES + c * 3] = plane(imgR, colbase - 1 + e, c);
    for (int e = threadIdx.x; e < nleft + 2; e += blockDim.x)
#pragma unroll
        for (int c = 0; c < CN; c++) lstage[e * ES + c * 3] = plane(imgL, clo + g.minX1 - 1 + e, c);
    if (KT == 0)
        for (int s = 0; s < K; s++) ring[s * 64 + lane] = 0;
    __syncthreads();
    auto finish = [&](uint32_t* dst, int e, int col) {  // entry e >= 1 holds image column col
#pragma unroll
        for (int c = 0; c < CN; c++) {
            uint32_t u = dst[e * ES + c * 3], l = dst[(e - 1) * ES + c * 3], r = dst[(e + 1) * ES + c * 3];
            uint32_t ul = col > 0 ? pk_lshr_u16(pk_add_u16(u, l), 0x00010001u) : u;
            uint32_t ur = col < g.W - 1 ? pk_lshr_u16(pk_add_u16(u, r), 0x00010001u) : u;
            dst[e * ES + c * 3 + 1] = pk_min_u16(pk_min_u16(ul, ur), u);
            dst[e * ES + c * 3 + 2] = pk_max_u16(pk_max_u16(ul, ur), u);
        }
    };
    for (int i = threadIdx.x; i < ncols; i += blockDim.x) finish(stage, i + 1, colbase + i);
    for (int i = threadIdx.x; i < nleft; i += blockDim.x) finish(lstage, i + 1, clo + g.minX1 + i);
    __syncthreads();
    const uint4* lent = reinterpret_cast<const uint4*>(lstage) + (ES / 4);  // skip the halo entry

    uint16_t* __restrict__ out = Hs + (size_t)pair * vol_stride + ((size_t)y * g.W1) * g.Dp + d;
    if (d >= g.Dp) return;  // lanes beyond the padded range only helped with the staging (no barrier follows)
    const uint32_t vmask = d < g.D ? 0xffffu : 0u;  // padded disparities D <= d < Dp are written as 0
    const uint4* ent = reinterpret_cast<const uint4*>(stage) + (size_t)(last - d + 1) * (ES / 4);

    // Software pipeline: the operands of step t+1 (three ds_read_b128 + the scalar loads of the left
    // pixel) are issued before the arithmetic of step t; two operand sets alternate (loop unrolled by 2).
    struct Ops { uint32_t V[CN], V0[CN], V1[CN], U[CN], U0[CN], U1[CN]; };
    auto fetch = [&](int t, Ops& o) {
        const int ct = min(max(t, 0), g.W1 - 1);  // clamped virtual column (box sum replicates the border)
        const uint4* e = ent + (size_t)(ct - clo) * (ES / 4);
        if (CN == 1) {
            uint4 a = e[0];
            o.V[0] = a.x; o.V0[0] = a.y; o.V1[0] = a.z;
        } else {
            uint4 a = e[0], b = e[1], c = e[2];
            o.V[0] = a.x; o.V0[0] = a.y; o.V1[0] = a.z;
            o.V[1 % CN] = a.w; o.V0[1 % CN] = b.x; o.V1[1 % CN] = b.y;
            o.V[2 % CN] = b.z; o.V0[2 % CN] = b.w; o.V1[2 % CN] = c.x;
        }
        const uint4* q = lent + (size_t)(ct - clo) * (ES / 4);
        if (CN == 1) {
            uint4 a = q[0];
            o.U[0] = a.x; o.U0[0] = a.y; o.U1[0] = a.z;
        } else {
            uint4 a = q[0], b = q[1], c = q[2];
            o.U[0] = a.x; o.U0[0] = a.y; o.U1[0] = a.z;
            o.U[1 % CN] = a.w; o.U0[1 % CN] = b.x; o.U1[1 % CN] = b.y;
            o.U[2 % CN] = b.z; o.U0[2 % CN] = b.w; o.U1[2 % CN] = c.x;
        }
    };
    uint32_t run = 0;
    const int t0 = xs - g.SW2, t1 = xe - 1 + g.SW2;
    // cost of one column for this lane's disparity (operands already fetched)
    auto column_cost = [&](const Ops& o) -> uint32_t {
        uint32_t acc = 0;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            // c0 = max(0, u - v1, v0 - u), c1 = max(0, v - u1, u0 - v): at most one term of each pair is
            // non-zero, so OR of the saturating differences is their max
            uint32_t a = pk_subsat_u16(o.U[c], o.V1[c]) | pk_subsat_u16(o.V0[c], o.U[c]);
            uint32_t b = pk_subsat_u16(o.V[c], o.U1[c]) | pk_subsat_u16(o.U0[c], o.V[c]);
            uint32_t m = pk_min_u16(a, b);
            m = pk_lshr_u16(m, 0x00020000u);  // raw plane: cost >> 2
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, m),
                                         __builtin_bit_cast(u16x2_t, 0x00010001u), acc, false);
        }
        return acc;
    };
    auto emit = [&](int t) {
        const int xo = t - g.SW2;
        if (xo >= xs)
            out[(size_t)xo * g.Dp] = (uint16_t)(run & vmask);  // one unconditional store: no exec juggling
    };
    Ops A, B;
    fetch(t0, A);
    if (KT > 0) {
        uint32_t rr[KT > 0 ? KT : 1];
#pragma unroll
        for (int j = 0; j < (KT > 0 ? KT : 1); j++) rr[j] = 0;
        for (int t = t0; t <= t1; t += 2 * KT) {
#pragma unroll
            for (int j = 0; j < 2 * KT; j++) {
                if (t + j <= t1) {  // uniform
                    const Ops& o = (j & 1) ? B : A;
                    fetch(min(t + j + 1, t1), (j & 1) ? A : B);
                    const uint32_t acc = column_cost(o);
                    run += acc - rr[j % (KT > 0 ? KT : 1)];
                    rr[j % (KT > 0 ? KT : 1)] = acc;
                    emit(t + j);
                }
            }
        }
    } else {
        int slot = 0;
        auto step = [&](int t, const Ops& o, Ops& nxt) {
            fetch(min(t + 1, t1), nxt);
            const uint32_t old = ring[slot * 64 + lane];
            const uint32_t acc = column_cost(o);
            ring[slot * 64 + lane] = acc;
            slot = slot + 1 == K ? 0 : slot + 1;
            run += acc - old;
            emit(t);
        };
        for (int t = t0; t <= t1; t += 2) {
            step(t, A, B);
            if (t + 1 <= t1) step(t + 1, B, A);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_vsum: C[y][x][d] = P2 + sum_{dy=-SH2..SH2} Hs[clamp(y+dy,0,H-1)][x][d]   (u16 wrap)
// One thread = 8 consecutive d (16 bytes) of one column, walking VSUM_ROWS rows downwards with a running
// sum: C(y) = C(y-1) + Hs(y+SH2) - Hs(y-SH2-1).  The K = 2*SH2+1 rows inside the window live in a
// thread-private LDS ring, so every Hs row is read once per row segment (plus K-1 halo rows) whatever the
// row pitch is -- a per-row kernel that re-reads its K rows only gets them from L2 when vertically
// adjacent workgroups happen to land on the same XCD (true for W1 = 1792, false for W1 = 1793).
// ------------------------------------------------------------------------------------------------
static constexpr int VSUM_ROWS = 64;

// SAT: the saturating recurrence of OpenCV's CV_SIMD build (see sgbm_cost.hpp); then one block walks ALL rows.
template <bool SAT>
__global__ __launch_bounds__(256) void k_vsum(const uint4* __restrict__ Hs, uint4* __restrict__ C, Geom g,
                                              size_t vol_stride16, int rows_per_block)
{
    extern __shared__ uint4 vring[];  // [K][256]
    const size_t rowv = (size_t)g.W1 * (g.Dp / 8);  // uint4 per row
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i0 < rowv;
    const size_t i = ok ? i0 : rowv - 1;
    const bool first_col = i < (size_t)(g.Dp / 8);  // cost column 0
    const int pair = blockIdx.z, H = g.H, SH2 = g.SW2, K = 2 * SH2 + 1;
    const int y0 = blockIdx.y * rows_per_block, y1 = min(y0 + rows_per_block, H);
    const uint4* base = Hs + (size_t)pair * vol_stride16 + i;
    uint4* out = C + (size_t)pair * vol_stride16 + i;
    auto ld = [&](int yy) -> uint4 { return base[(size_t)min(max(yy, 0), H - 1) * rowv]; };
    auto add = [&](uint32_t a, uint32_t b) { return SAT ? pk_addsat_i16(a, b) : pk_add_u16(a, b); };
    auto upd = [&](uint32_t a, uint32_t v, uint32_t o, bool add_first) {
        if (!SAT) return pk_sub_u16(pk_add_u16(a, v), o);
        return add_first ? pk_subsat_i16(pk_addsat_i16(a, v), o) : pk_addsat_i16(pk_subsat_i16(a, o), v);
    };
    const uint32_t p2 = dup16((uint32_t)g.P2);
    uint4 acc = make_uint4(p2, p2, p2, p2);
    for (int j = 0; j < K; j++) {
        uint4 v = ld(y0 - SH2 + j);
        vring[j * 256 + threadIdx.x] = v;
        acc.x = add(acc.x, v.x); acc.y = add(acc.y, v.y);
        acc.z = add(acc.z, v.z); acc.w = add(acc.w, v.w);
    }
    if (ok) out[(size_t)y0 * rowv] = acc;
    int slot = 0;  // ring position of the oldest row (y - SH2 - 1 of the next output row)
    uint4 nx[4];   // rows y+SH2 .. y+3+SH2 in flight
#pragma unroll
    for (int u = 0; u < 4; u++) nx[u] = ld(y0 + 1 + u + SH2);
    for (int y = y0 + 1; y < y1; y += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (y + u < y1) {
                const uint4 v = nx[u];
                nx[u] = ld(y + u + 4 + SH2);
                const uint4 o = vring[slot * 256 + threadIdx.x];
                vring[slot * 256 + threadIdx.x] = v;
                slot = slot + 1 == K ? 0 : slot + 1;
                const bool af = first_col && y + u + SH2 < H;  // OpenCV's order in column 0 while the entering row exists
                acc.x = upd(acc.x, v.x, o.x, af); acc.y = upd(acc.y, v.y, o.y, af);
                acc.z = upd(acc.z, v.z, o.z, af); acc.w = upd(acc.w, v.w, o.w, af);
                if (ok) out[(size_t)(y + u) * rowv] = acc;
            }
        }
    }
}

// Register-ring variant for the common block sizes (K = 2*SH2+1 known at compile time): no LDS at all, so
// its workgroups can share a CU with the LDS-hungry cost kernel of another stream.
template <int K>
__global__ __launch_bounds__(256) void k_vsum_reg(const uint4* __restrict__ Hs, uint4* __restrict__ C, Geom g,
                                                  size_t vol_stride16)
{
    constexpr int SH2 = K / 2;
    const size_t rowv = (size_t)g.W1 * (g.Dp / 8);
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i0 < rowv;
    const size_t i = ok ? i0 : rowv - 1;
    const int pair = blockIdx.z, H = g.H;
    const int y0 = blockIdx.y * VSUM_ROWS, y1 = min(y0 + VSUM_ROWS, H);
    const uint4* base = Hs + (size_t)pair * vol_stride16 + i;
    uint4* out = C + (size_t)pair * vol_stride16 + i;
    auto ld = [&](int yy) -> uint4 { return base[(size_t)min(max(yy, 0), H - 1) * rowv]; };
    const uint32_t p2 = dup16((uint32_t)g.P2);
    uint4 acc = make_uint4(p2, p2, p2, p2);
    uint4 ring[K];  // slot j: row y0 - SH2 + j, later replaced in rotation (static indices: the loop steps by K)
#pragma unroll
    for (int j = 0; j < K; j++) {
        ring[j] = ld(y0 - SH2 + j);
        acc.x = pk_add_u16(acc.x, ring[j].x); acc.y = pk_add_u16(acc.y, ring[j].y);
        acc.z = pk_add_u16(acc.z, ring[j].z); acc.w = pk_add_u16(acc.w, ring[j].w);
    }
    if (ok) out[(size_t)y0 * rowv] = acc;
    for (int y = y0 + 1; y < y1; y += K) {
        uint4 nv[K];
#pragma unroll
        for (int j = 0; j < K; j++) nv[j] = ld(y + j + SH2);  // unconditional (clamped): issued back to back
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint4 v = nv[j], o = ring[j];
            ring[j] = v;
            acc.x = pk_sub_u16(pk_add_u16(acc.x, v.x), o.x); acc.y = pk_sub_u16(pk_add_u16(acc.y, v.y), o.y);
            acc.z = pk_sub_u16(pk_add_u16(acc.z, v.z), o.z); acc.w = pk_sub_u16(pk_add_u16(acc.w, v.w), o.w);
            if (ok && y + j < y1) out[(size_t)(y + j) * rowv] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan: L_r along direction r = (dx, dy) for every line of the cost array, accumulated into S.
//   L(p,d) = C(p,d) + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, minLp+P2) - (minLp+P2),  Lp = L(p-r,.)
//   Lp = 0, minLp = 0 outside the array; Lp[-1] = Lp[D] = MAX_COST.
// A line is owned by LANES lanes; lane l holds d in [l*2*NR, (l+1)*2*NR) as NR packed u16 pairs.
// All arithmetic is u16: real values are in [0, 32767], MAX_COST + P1 does not wrap, and the final
// (C + m) - delta is exact modulo 2^16 (OpenCV's (CostType) cast).
// ------------------------------------------------------------------------------------------------
// directions of one launch: blockIdx.z selects the entry; with more than one entry every direction
// writes its own volume (Sv + z * dir_stride, FIRST only) so that all of them run concurrently
// ---- a lane's slice of a pixel's disparity vector: NR packed registers = 2*NR consecutive disparities = 4*NR bytes,
// 4-byte aligned (16-byte aligned, and moved as uint4, when NR % 4 == 0).  In LDS a lane's slice takes NQ = ceil(NR/4)
// 16-byte slots, the tail zero (lds_ld_regs / lds_st_regs below).
template <int NR> struct __attribute__((packed, aligned(4))) RegVec { uint32_t v[NR]; };
template <int NR>
__device__ __forceinline__ void ld_regs(const uint16_t* __restrict__ p, uint32_t (&dst)[NR])
{
    if constexpr (NR % 4 == 0) {
        const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int v = 0; v < NR / 4; v++) {
            const uint4 w = q[v];
            dst[4 * v] = w.x; dst[4 * v + 1] = w.y; dst[4 * v + 2] = w.z; dst[4 * v + 3] = w.w;
        }
    } else {
        RegVec<NR> t;
        __builtin_memcpy(&t, p, sizeof(t));
#pragma unroll
        for (int k = 0; k < NR; k++) dst[k] = t.v[k];
    }
}
template <int NR>
__device__ __forceinline__ void st_regs(uint16_t* __restrict__ p, const uint32_t (&src)[NR])
{
    if constexpr (NR % 4 == 0) {
        uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
        for (int v = 0; v < NR / 4; v++) q[v] = make_uint4(src[4 * v], src[4 * v + 1], src[4 * v + 2], src[4 * v + 3]);
    } else {
        RegVec<NR> t;
#pragma unroll
        for (int k = 0; k < NR; k++) t.v[k] = src[k];
        __builtin_memcpy(p, &t, sizeof(t));
    }
}
// streaming variants (global_load / global_store ... nt): the volumes are read and written once per pass.  Measured on the
// band passes (CAMD_BAND_NT, tools/history/gpu_r6_nt.sh) -- see sgbm_band.hpp
typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
template <int NR>
__device__ __forceinline__ void ld_regs_nt(const uint16_t* __restrict__ p, uint32_t (&dst)[NR])
{
    if constexpr (NR % 4 == 0) {
        const nt_u32x4* q = reinterpret_cast<const nt_u32x4*>(p);
#pragma unroll
        for (int v = 0; v < NR / 4; v++) {
            const nt_u32x4 w = __builtin_nontemporal_load(q + v);
            dst[4 * v] = w.x; dst[4 * v + 1] = w.y; dst[4 * v + 2] = w.z; dst[4 * v + 3] = w.w;
        }
    } else {
        ld_regs<NR>(p, dst);
    }
}
template <int NR>
__device__ __forceinline__ void st_regs_nt(uint16_t* __restrict__ p, const uint32_t (&src)[NR])
{
    if constexpr (NR % 4 == 0) {
        nt_u32x4* q = reinterpret_cast<nt_u32x4*>(p);
#pragma unroll
        for (int v = 0; v < NR / 4; v++)
            __builtin_nontemporal_store(nt_u32x4{src[4 * v], src[4 * v + 1], src[4 * v + 2], src[4 * v + 3]}, q + v);
    } else {
        st_regs<NR>(p, src);
    }
}
// The line-scan kernels and k_wta (the latency path) with `nt` loads: CAMD_SCAN_NT 1.  Measured no faster, rather slower
// (one 1080p pair 1.79-1.88 -> 1.89-1.90 ms; MODE_HH 2.59-2.64 -> 2.70-2.91): there the five or eight direction scans of
// ONE pair read the same C concurrently, and the L2 / MALL reuse that `nt` gives up is worth having.  Off.
#ifndef CAMD_SCAN_NT
#define CAMD_SCAN_NT 0
#endif
#if CAMD_SCAN_NT
#define CAMD_SCAN_LD(p, dst) ld_regs_nt<NR>(p, dst)
#else
#define CAMD_SCAN_LD(p, dst) ld_regs<NR>(p, dst)
#endif
template <int NR> __device__ __forceinline__ uint32_t reg_or0(const uint32_t (&a)[NR], int i) { return i < NR ? a[i < NR ? i : 0] : 0u; }
// LDS: slot v of lane `idx` lives at p[v * stride + idx] -- one PLANE per slot, so that consecutive lanes are 16 bytes
// apart in every ds_read_b128 / ds_write_b128 (lane-major slots, p[idx * NQ + v], put the lanes 32 bytes apart at
// NQ = 2: two-way bank conflicts in every exchange of the D > 128 band passes, 44 % of their LDS cycles in round 4)
template <int NR>
__device__ __forceinline__ void lds_ld_regs(const uint4* p, int idx, int stride, uint32_t (&dst)[NR])
{
#pragma unroll
    for (int v = 0; v < (NR + 3) / 4; v++) {
        const uint4 w = p[v * stride + idx];
        const uint32_t e[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (4 * v + k < NR) dst[4 * v + k] = e[k];
    }
}
template <int NR>
__device__ __forceinline__ void lds_st_regs(uint4* p, int idx, int stride, const uint32_t (&src)[NR])
{
#pragma unroll
    for (int v = 0; v < (NR + 3) / 4; v++)
        p[v * stride + idx] = make_uint4(reg_or0<NR>(src, 4 * v), reg_or0<NR>(src, 4 * v + 1), reg_or0<NR>(src, 4 * v + 2),
                                         reg_or0<NR>(src, 4 * v + 3));
}

struct ScanDirs {
    int dx[8], dy[8], nlines[8];
    size_t dir_stride;
};

template <int LANES, int NR, bool FIRST, bool PAD>
__global__ __launch_bounds__(256) void k_scan(const uint16_t* __restrict__ Cv, uint16_t* __restrict__ Sbase,
                                              Geom g, ScanDirs sd, size_t vol_stride)
{
    const int dx = sd.dx[blockIdx.z], dy = sd.dy[blockIdx.z], nlines = sd.nlines[blockIdx.z];
    uint16_t* __restrict__ Sv = Sbase + (size_t)blockIdx.z * sd.dir_stride;
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int line = tid / LANES, li = tid % LANES;
    if (line >= nlines) return;
    const int pair = blockIdx.y;
    const int W1 = g.W1, H = g.H;

    // start pixel and length of this line
    int x0, y0;
    if (dy == 0) {
        y0 = line;
        x0 = dx > 0 ? 0 : W1 - 1;
    } else {
        const int ys = dy > 0 ? 0 : H - 1;
        if (dx == 0 || line < W1) {
            x0 = line;
            y0 = ys;
        } else {
            x0 = dx > 0 ? 0 : W1 - 1;
            int k = line - W1 + 1;  // 1..H-1
            y0 = dy > 0 ? k : H - 1 - k;
        }
    }
    int len;
    {
        int lx = dx == 0 ? (1 << 30) : (dx > 0 ? W1 - x0 : x0 + 1);
        int ly = dy == 0 ? (1 << 30) : (dy > 0 ? H - y0 : y0 + 1);
        len = min(lx, ly);
    }
    const size_t off = (size_t)pair * vol_stride + ((size_t)y0 * W1 + x0) * g.Dp + (size_t)li * (2 * NR);
    const ptrdiff_t step = ((ptrdiff_t)dy * W1 + dx) * (ptrdiff_t)g.Dp;
    const uint16_t* cp = Cv + off;
    uint16_t* sp = Sv + off;

    uint32_t keep[NR], sent[NR];
    if (PAD) {
#pragma unroll
        for (int k = 0; k < NR; k++) {
            int d0 = li * 2 * NR + 2 * k;
            uint32_t kp = (d0 < g.D ? 0xffffu : 0u) | (d0 + 1 < g.D ? 0xffff0000u : 0u);
            keep[k] = kp;
            sent[k] = ~kp & SENT_PK;
        }
    }

    const uint32_t P1pk = dup16((uint32_t)g.P1), P2pk = dup16((uint32_t)g.P2);
    uint32_t Lp[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) Lp[k] = 0;
    uint32_t delta = P2pk;  // minLp = 0
    uint32_t edge_lo = SENT_PK, edge_hi = SENT_PK;

    // A line is a dependent chain (every pixel needs the previous one), so with one pair per call the kernel is
    // latency-bound: the C (and S) vectors of the next PF-1 pixels are kept in flight in a register ring.  The
    // loop is unrolled by PF so the ring never moves, and the loads are unconditional (clamped to the line's last
    // pixel) so that the compiler can wait with counted vmcnt(N) instead of draining the ring every step.
    constexpr int PF = NR <= 4 ? 8 : (NR <= 8 ? 4 : 2);
    uint32_t cr[PF][NR], sr[FIRST ? 1 : PF][NR];
#pragma unroll
    for (int u = 0; u < PF - 1; u++) {
        const ptrdiff_t o = (ptrdiff_t)min(u, len - 1) * step;
        CAMD_SCAN_LD(cp + o, cr[u]);
        if (!FIRST) CAMD_SCAN_LD(sp + o, sr[u]);
    }
    for (int i0 = 0; i0 < len; i0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int i = i0 + u;
            {
                const ptrdiff_t o = (ptrdiff_t)min(i + PF - 1, len - 1) * step;
                CAMD_SCAN_LD(cp + o, cr[(u + PF - 1) % PF]);
                if (!FIRST) CAMD_SCAN_LD(sp + o, sr[(u + PF - 1) % PF]);
            }
            if (i < len) {
                const uint32_t(&c)[NR] = cr[u];
                // neighbours across lanes: d-1 of my first element, d+1 of my last element
                // (edge_lo / edge_hi persist: the lane a row shift leaves untouched keeps its MAX_COST sentinel)
                edge_lo = dpp_mov<DPP_ROW_SHR1>(edge_lo, Lp[NR - 1]);
                edge_hi = dpp_mov<DPP_ROW_SHL1>(edge_hi, Lp[0]);
                uint32_t prev_last = edge_lo, next_first = edge_hi;
                if (LANES < 16) {
                    if (li == 0) prev_last = SENT_PK;
                    if (li == LANES - 1) next_first = SENT_PK;
                }
                // m[k] = (Lp[2k-1], Lp[2k]) ; m[k+1] = (Lp[2k+1], Lp[2k+2])
                uint32_t m[NR + 1];
                m[0] = alignbit16(Lp[0], prev_last);
#pragma unroll
                for (int k = 1; k < NR; k++) m[k] = alignbit16(Lp[k], Lp[k - 1]);
                m[NR] = alignbit16(next_first, Lp[NR - 1]);

                uint32_t L[NR];
                uint32_t mn = SENT_PK;
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    // C + min(Lp, t, delta) - delta  ==  C - max(delta - min(Lp, t), 0)   (mod 2^16): one op less
                    uint32_t t = pk_add_u16(pk_min_u16(m[k], m[k + 1]), P1pk);
                    uint32_t l = pk_sub_u16(c[k], pk_subsat_u16(delta, pk_min_u16(Lp[k], t)));
                    if (PAD) l = (l & keep[k]) | sent[k];
                    L[k] = l;
                    mn = pk_min_u16(mn, l);
                }
                mn = group_min_pk_u16<LANES>(mn);
                mn = pk_min_u16(mn, alignbit16(mn, mn));  // both halves = min over all d
                delta = pk_add_u16(mn, P2pk);

                uint32_t s[NR];
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    s[k] = FIRST ? L[k] : pk_addsat_i16(sr[FIRST ? 0 : u][k], L[k]);
                    Lp[k] = L[k];
                }
                st_regs<NR>(sp + (ptrdiff_t)i * step, s);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// k_wta: one workgroup per image row.  Per cost column x (a LANES-lane group each):
//   minS / bestDisp (smallest d attaining it), uniqueness test, sub-pixel parabola,
//   right-view map disp2 by LDS atomicMin on (minS << 16 | 0xFFFF - d)  [ties keep the larger d,
//   i.e. the larger x, which OpenCV visits first], then the left-right check of the row.
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t KEY_INIT = 0x7fff0000u;

// EXACT (sgbm_exact.hpp): Sv points at nvol per-direction volumes of int L values (before narrowing); they are added
// up in OpenCV's grouping and with each mode's own narrowing -- combine 0: saturate(L0 + L1 + L2 + L3) of the int
// values, then saturate(that + the rest) (computeDisparitySGBM); 1: one saturating add per volume, in order, of
// (CostType)L (computeDisparitySGBM_HH4); 2: the same of saturate(L) (the 3-way loop) -- and every total is carried
// as S + 32768 in an unsigned half, so that all comparisons below order the same way; `bias` turns them back into
// values where the arithmetic needs them.
// (the body of k_wta for row y of pair `pair`; the persistent exact kernel of sgbm_exact.hpp calls it row after row)
template <int LANES, int NR, bool EXACT>
__device__ __forceinline__ void wta_row(const uint16_t* __restrict__ Sv, int16_t* __restrict__ disp,
                                        size_t disp_pitch_e, size_t disp_stride_e, const Geom& g,
                                        size_t vol_stride, int nvol, size_t dir_stride, int tie_lanes,
                                        int combine, int y, int pair)
{
    constexpr int bias = EXACT ? 32768 : 0;
    constexpr uint32_t key_init = EXACT ? 0xffff0000u : KEY_INIT;
    constexpr int max_cost_b = MAX_COST + bias;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);          // [W]
    int16_t* d1row = reinterpret_cast<int16_t*>(keys + g.W);     // [W]
    constexpr int GROUPS = 256 / LANES;
    const int li = threadIdx.x % LANES, grp = threadIdx.x / LANES;
    const int INVALID_SCALED = (g.minD - 1) * 16;

    for (int x = threadIdx.x; x < g.W; x += 256) {
        keys[x] = key_init;
        d1row[x] = (int16_t)INVALID_SCALED;
    }
    __syncthreads();

    const uint16_t* Srow = Sv + (size_t)pair * vol_stride + ((size_t)y * g.W1) * g.Dp + (size_t)li * (2 * NR);
    const int dbase = li * 2 * NR;
    for (int x = grp; x < g.W1; x += GROUPS) {
        uint32_t s[NR];
        if (!EXACT) {
            CAMD_SCAN_LD(Srow + (size_t)x * g.Dp, s);
            // concurrent-direction path: S = saturating sum of the per-direction volumes
            for (int dv = 1; dv < nvol; dv++) {
                uint32_t q[NR];
                CAMD_SCAN_LD(Srow + (size_t)dv * dir_stride + (size_t)x * g.Dp, q);
#pragma unroll
                for (int k = 0; k < NR; k++) s[k] = pk_addsat_i16(s[k], q[k]);
            }
        } else {
            // (dir_stride and the row offsets count int elements here)
            const int32_t* Lrow = reinterpret_cast<const int32_t*>(Sv) + ((size_t)y * g.W1 + x) * g.Dp + (size_t)li * (2 * NR);
            int tot[2 * NR], part[2 * NR];
#pragma unroll
            for (int e = 0; e < 2 * NR; e++) tot[e] = part[e] = 0;
            for (int dv = 0; dv <= nvol; dv++) {
                // close the running group before volume dv joins: after every volume in the sequential modes, at the
                // group boundary (after the first four volumes) and at the end in combine 0
                if (dv > 0 && (combine != 0 || dv == 4 || dv == nvol)) {
#pragma unroll
                    for (int e = 0; e < 2 * NR; e++) {
                        const int t = tot[e] + part[e];
                        tot[e] = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                        part[e] = 0;
                    }
                }
                if (dv == nvol) break;
                const int2* pv = reinterpret_cast<const int2*>(Lrow + (size_t)dv * dir_stride);  // 8-byte aligned: li * 2NR ints
#pragma unroll
                for (int v = 0; v < NR; v++) {
                    const int2 q = pv[v];
                    const int w[2] = {q.x, q.y};
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const int L = w[k];
                        part[2 * v + k] += combine == 0 ? L : (combine == 1 ? (int)(int16_t)L
                                                                             : (L < -32768 ? -32768 : (L > 32767 ? 32767 : L)));
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NR; k++)
                s[k] = (uint32_t)(tot[2 * k] + bias) | ((uint32_t)(tot[2 * k + 1] + bias) << 16);
        }
        // (S << 16 | d) minimum: smallest S, then smallest d
        uint32_t key = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < NR; k++) {
            int d0 = dbase + 2 * k;
            uint32_t lo = s[k] & 0xffffu, hi = s[k] >> 16;
            if (d0 < g.D) key = min(key, (lo << 16) | (uint32_t)d0);
            if (d0 + 1 < g.D) key = min(key, (hi << 16) | (uint32_t)(d0 + 1));
        }
        key = group_min_u32<LANES>(key);
        int minS = (int)(key >> 16), best = (int)(key & 0xffffu);
        if (NR % 4 == 0 && tie_lanes == 8) {  // (MODE_SGBM_3WAY keeps layouts of whole groups of 8 per lane: normalise())
            // MODE_SGBM_3WAY as OpenCV's CV_SIMD build decides ties (oracle/sgbm_ref.c way3_winner): the disparities
            // below E are scanned 8 at a time, every one of the 8 lane slots keeps the LAST d that attains its minimum,
            // the winner is the smallest of those positions among the slots that hold the global minimum; the scalar
            // tail [E, D) only wins with a strictly smaller total.  A lane owns whole groups of 8 consecutive d here, so
            // element e of every group is slot e.
            const int E = (g.D % 8 == 0) ? g.D : 8 * ((g.D - 1) / 8);
            uint32_t m1 = 0xffffu, ktail = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < NR; k++) {
                const int d0 = dbase + 2 * k;
                const uint32_t lo = s[k] & 0xffffu, hi = s[k] >> 16;
                if (d0 < E) m1 = min(m1, lo); else if (d0 < g.D) ktail = min(ktail, (lo << 16) | (uint32_t)d0);
                if (d0 + 1 < E) m1 = min(m1, hi); else if (d0 + 1 < g.D) ktail = min(ktail, (hi << 16) | (uint32_t)(d0 + 1));
            }
            m1 = group_min_u32<LANES>(m1);
            ktail = group_min_u32<LANES>(ktail);
            uint32_t pos = 0xffffffffu;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                uint32_t last = 0;  // 1 + the largest d of slot e (in this lane) whose total is the minimum
#pragma unroll
                for (int v = 0; v < NR / 4; v++) {
                    const int k = 4 * v + e / 2, d = dbase + 8 * v + e;
                    const uint32_t val = (e & 1) ? (s[k] >> 16) : (s[k] & 0xffffu);
                    if (d < E && val == m1) last = (uint32_t)d + 1;
                }
                last = group_max_u32<LANES>(last);
                if (last) pos = min(pos, last - 1);
            }
            if (E > 0 && (ktail >> 16) >= m1) { minS = (int)m1; best = (int)pos; }
            else { minS = (int)(ktail >> 16); best = (int)(ktail & 0xffffu); }
        }
        // uniqueness + the neighbours of the winner
        uint32_t flags = 0, sm = 0, spv = 0;
        const int thr = (minS - bias) * 100, mul = 100 - g.uniq;
#pragma unroll
        for (int k = 0; k < NR; k++) {
            int d0 = dbase + 2 * k;
            int lo = (int)(s[k] & 0xffffu) - bias, hi = (int)(s[k] >> 16) - bias;
            if (d0 < g.D) {
                if (lo * mul < thr && abs(best - d0) > 1) flags = 1;
                if (d0 == best - 1) sm = s[k] & 0xffffu;
                if (d0 == best + 1) spv = s[k] & 0xffffu;
            }
            if (d0 + 1 < g.D) {
                if (hi * mul < thr && abs(best - d0 - 1) > 1) flags = 1;
                if (d0 + 1 == best - 1) sm = s[k] >> 16;
                if (d0 + 1 == best + 1) spv = s[k] >> 16;
            }
        }
        // the two neighbours as 16-bit fields (each set by at most one lane; a field left at zero is never used)
        const uint32_t packed = group_or_u32<LANES>((sm << 16) | spv);
        flags = group_or_u32<LANES>(flags);
        if (li == 0 && minS < max_cost_b && !flags) {
            const int Sm = (int)(packed >> 16) - bias, Sp = (int)(packed & 0xffffu) - bias;
            int d = best;
            int x2 = x + g.minX1 - d - g.minD;
            atomicMin(&keys[x2], ((uint32_t)minS << 16) | (uint32_t)(0xffff - d));
            minS -= bias;
            if (0 < d && d < g.D - 1) {
                int denom2 = max(Sm + Sp - 2 * minS, 1);
                d = d * 16 + ((Sm - Sp) * 16 + denom2) / (denom2 * 2);  // C division truncates
            } else
                d *= 16;
            d1row[x + g.minX1] = (int16_t)(d + g.minD * 16);
        }
    }
    __syncthreads();

    int16_t* out = disp + (size_t)pair * disp_stride_e + (size_t)y * disp_pitch_e;
    const int maxX1 = g.minX1 + g.W1;
    for (int x = threadIdx.x; x < g.W; x += 256) {
        int d1 = d1row[x];
        if (x >= g.minX1 && x < maxX1 && d1 != INVALID_SCALED) {
            int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            int _x = x - _d, x_ = x - d_;
            bool bad = true;
            if (0 <= _x && _x < g.W) {
                uint32_t k = keys[_x];
                // untouched entries hold INVALID_DISP_SCALED and are compared unscaled (OpenCV quirk)
                int v = k == key_init ? INVALID_SCALED : (int)(0xffffu - (k & 0xffffu)) + g.minD;
                bad = v >= g.minD && abs(v - _d) > g.d12;
            } else
                bad = false;
            if (bad) {
                if (0 <= x_ && x_ < g.W) {
                    uint32_t k = keys[x_];
                    int v = k == key_init ? INVALID_SCALED : (int)(0xffffu - (k & 0xffffu)) + g.minD;
                    bad = v >= g.minD && abs(v - d_) > g.d12;
                } else
                    bad = false;
            }
            if (bad) d1 = INVALID_SCALED;
        }
        out[x] = (int16_t)d1;
    }
}

template <int LANES, int NR>
__global__ __launch_bounds__(256) void k_wta(const uint16_t* __restrict__ Sv, int16_t* __restrict__ disp,
                                             size_t disp_pitch_e, size_t disp_stride_e, Geom g,
                                             size_t vol_stride, int nvol, size_t dir_stride, int tie_lanes)
{
    wta_row<LANES, NR, false>(Sv, disp, disp_pitch_e, disp_stride_e, g, vol_stride, nvol, dir_stride, tie_lanes, 0,
                              (int)blockIdx.x, (int)blockIdx.y);
}

__global__ void k_fill_s16(int16_t* p, size_t pitch_e, size_t stride_e, int W, int H, int value)
{
    int x = blockIdx.x * 256 + threadIdx.x;
    if (x < W) p[(size_t)blockIdx.z * stride_e + (size_t)blockIdx.y * pitch_e + x] = (int16_t)value;
}

}  // namespace camd
#include "sgbm_band.hpp"
#include "sgbm_cost.hpp"
#include "sgbm_exact.hpp"
namespace camd {

// MODE_SGBM_3WAY: rows of the final raw disparity come from the stripe that owns them
__global__ __launch_bounds__(256) void k_gather_stripes(const int16_t* __restrict__ rawv, size_t rawv_stride_e,
                                                        int16_t* __restrict__ raw, size_t raw_stride_e, int W, int H,
                                                        int stripe_sz, CostRanges cr, const uint32_t* __restrict__ err,
                                                        const uint32_t* __restrict__ refused, int invalid)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pair = blockIdx.z;
    if (x >= W) return;
    const int s = min(y / stripe_sz, cr.n - 1);
    // a band pass that gave up waiting (sgbm_band.hpp) must not hand back plausible garbage: like k_lrcheck (bit 0 of
    // the error word); and a pair with ANY refused stripe (`refused`: the per-volume below-P2 flags, passed only when
    // there is no exact path to redo them) is invalid as a whole, as the header promises -- not just that stripe's rows
    bool bad = err && (*err & 1u);
    if (refused)
        for (int k = 0; k < cr.n; k++) bad |= refused[pair * cr.n + k] != 0;
    raw[(size_t)pair * raw_stride_e + (size_t)y * W + x] =
        bad ? (int16_t)invalid : rawv[(size_t)(pair * cr.n + s) * rawv_stride_e + (size_t)(y - cr.start[s]) * W + x];
}


// defined in post.hip
int launch_median3(const int16_t* src, size_t src_pitch_e, size_t src_stride_e, int16_t* dst,
                   size_t dst_pitch_e, size_t dst_stride_e, int w, int h, int batch, hipStream_t st);
int launch_speckle(int16_t* img, size_t pitch_e, size_t stride_e, int w, int h, int new_val, int max_size,
                   int max_diff, void* ws, size_t ws_bytes, int batch, hipStream_t st, bool* clean);
size_t speckle_ws_bytes(int w, int h, int batch);

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
enum Stage { ST_COST = 0, ST_HSUM, ST_VSUM, ST_SCAN, ST_SCAN2, ST_WTA, ST_POST, ST_SPECKLE, ST_COUNT };
// "cost" = the fused cost kernel (then hsum / vsum are empty), "hsum" + "vsum" = the split pair (then cost is empty);
// "scan" = the aggregation launches; on the band path the last pass is timed separately as "scan_last"
static const char* kStageNames[ST_COUNT] = {"cost", "hsum", "vsum", "scan", "scan_last", "wta", "median", "speckle"};

}  // namespace camd

struct camd_sgbm {
    camd::Geom g;         // the image
    camd::Geom ga;        // what the aggregation kernels see: g, or (MODE_SGBM_3WAY) one stripe of at most ga.H rows
    camd::CostRanges cr;  // row ranges of the cost volume: the image, or the 3WAY stripes (each a "virtual pair")
    int stripe_sz;        // 3WAY: rows a stripe owns
    int16_t* rawv;        // 3WAY: raw disparity per virtual pair [max_batch * cr.n][ga.H][W]
    uint32_t* cost_ovf;   // per volume: the wrapping cost kernel saw a value too close to 32767 (two-stage saturating build)
    uint32_t* cost_neg;   // per volume: C holds a value below P2 -> outside the packed-u16 regime (sgbm_exact.hpp)
    bool may_overflow;    // the parameters allow an int16 overflow of the box sums at all
    int exact_cap;        // 1: flagged volumes take the exact path (Lx allocated, CAMD_OPT_EXACT on); 0: they are refused
    int32_t* Lx;          // exact path: npaths per-direction volumes of int L values for ONE flagged volume
    int way3_simd_lanes;  // 3WAY winner-take-all tie rule: 8 = cv2's SSE / NEON builds (default), 1 = scalar build
    camd_sgbm_params params;
    int max_batch;
    size_t vol_elems;     // per pair, int16 elements of one volume
    uint16_t *C, *S;      // S doubles as the hsum buffer before aggregation
    int16_t* raw;         // [max_batch][H][W] disparity before median
    void* speckle_ws;
    bool speckle_clean;   // every parent entry of speckle_ws is -1 (post.hip keeps it so from call to call)
    hipStream_t speckle_stream;  // ... by the kernels of the last call, which ran on this stream
    // band-wavefront path (sgbm_band.hpp)
    bool band_ok;         // geometry supported by k_band instantiations
    int path;             // 0 = band passes (default when band_ok), 1 = one k_scan per direction
    int keep_S;           // band path: also store S in the final pass (stage-wise parity hook)
    int cost_path;        // CAMD_COST_*
    int saturate;         // U7: 1 = C saturates like OpenCV's CV_SIMD build (default), 0 = wraps like the scalar build
    int phases;           // CAMD_OPT_PHASES (experimental): bit 0 = build the cost volume, bit 1 = first aggregation pass,
                          // bit 2 = last pass + winner-take-all + post filters (default 7 = everything)
    int nbands, nchunks;
    size_t erec_stride;
    unsigned long long* E;
    uint32_t *flags, *ticket, *err, *keys;
    uint32_t* xbar;       // grid-barrier counter of the exact path's persistent kernel (sgbm_exact.hpp)
    int num_cus;          // compute units of the device the handle lives on
    int persist_cost, persist_row;  // CAMD_OPT_RESIDENT (experimental): workgroups per CU of k_cost_persist / k_band_row_persist, 0 = off
    uint32_t* err_host;   // pinned mirror of *err, refreshed by an async copy after every band-path compute
    int16_t* d1;
    uint32_t epoch;
    uint16_t* Smulti;     // concurrent-direction path: npaths volumes for smulti_cap pairs (allocated in create / set_option)
    int smulti_cap;
    int last_batch;
    bool profiling;
    hipEvent_t ev[camd::ST_COUNT + 1];
    bool ev_ok;
};

namespace camd {

static int normalise(const camd_sgbm_params* p, int width, int height, int cn, Geom* g)
{
    if (!p) { set_error("params is NULL"); return CAMD_ERR_BAD_ARG; }
    if (width <= 0 || height <= 0 || (cn != 1 && cn != 3)) {
        set_error("need width, height > 0 and 1 or 3 channels (got %d x %d x %d)", width, height, cn);
        return CAMD_ERR_BAD_ARG;
    }
    if (p->numDisparities <= 0) { set_error("numDisparities must be > 0"); return CAMD_ERR_BAD_ARG; }
    if (p->mode < CAMD_MODE_SGBM || p->mode > CAMD_MODE_HH4) {
        set_error("mode %d unknown (MODE_SGBM=0, MODE_HH=1, MODE_SGBM_3WAY=2, MODE_HH4=3)", p->mode);
        return CAMD_ERR_BAD_ARG;
    }
    memset(g, 0, sizeof(*g));
    g->W = width; g->H = height; g->cn = cn;
    g->minD = p->minDisparity; g->D = p->numDisparities;
    int maxD = g->minD + g->D;
    g->minX1 = maxD > 0 ? maxD : 0;
    int maxX1 = width + (g->minD < 0 ? g->minD : 0);
    g->W1 = maxX1 - g->minX1;
    g->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    g->uniq_magic = (g->uniq <= 98) ? (uint32_t)((1ull << 32) / (unsigned)(100 - g->uniq) + 1) : 0u;
    g->d12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    g->P1 = p->P1 > 0 ? p->P1 : 2;
    int P2 = p->P2 > 0 ? p->P2 : 5;
    g->P2 = P2 > g->P1 + 1 ? P2 : g->P1 + 1;
    int bs = p->blockSize > 0 ? p->blockSize : 5;
    g->SW2 = bs / 2;
    if (p->mode == CAMD_MODE_SGBM_3WAY && p->blockSize <= 0) g->SW2 = 1;  // cv2's 3-way loop: SADWindowSize > 0 ? /2 : 1
    g->ftzero = (p->preFilterCap > 15 ? p->preFilterCap : 15) | 1;
    g->mode = p->mode;
    g->npaths = p->mode == CAMD_MODE_HH ? 8 : (p->mode == CAMD_MODE_HH4 ? 4 : (p->mode == CAMD_MODE_SGBM_3WAY ? 3 : 5));
    g->speckleWindowSize = p->speckleWindowSize;
    g->speckleRange = p->speckleRange;
    if (2 * g->SW2 + 1 > HSUM_RING) {
        set_error("blockSize %d > %d not implemented", bs, HSUM_RING - 1);
        return CAMD_ERR_UNSUPPORTED;
    }
    // (P2: cv2's rule of thumb 32 * cn * blockSize^2 is 21600 at block 15 RGB; the fuzz covers the range up to the
    // limit -- in the last few per cent below 32767 the exact int path meets narrowings it does not restate)
    if (g->P2 > CAMD_MAX_P2 || g->ftzero > CAMD_MAX_FTZERO) {
        set_error("P2 = %d / preFilterCap = %d outside the int16 regime the kernels implement", g->P2,
                  p->preFilterCap);
        return CAMD_ERR_UNSUPPORTED;
    }
    if (g->W1 > 0 && g->W1 <= g->SW2) {
        // cv2's first box sum reads pixel-cost columns 0..SW2 without clamping to width1-1: with fewer
        // columns than that it reads memory it never wrote, so there is no reference answer to match
        set_error("only %d matchable columns for blockSize %d: cv2.StereoSGBM's result is undefined there "
                  "(needs width - numDisparities > blockSize / 2)", g->W1, bs);
        return CAMD_ERR_UNSUPPORTED;
    }
    if (g->D > 512) { set_error("numDisparities %d > 512 not implemented", g->D); return CAMD_ERR_UNSUPPORTED; }
    // Layout of a pixel's disparity vector: `lanes` lanes x `nr` packed registers (2 disparities each).  With 16 lanes a
    // register more per lane is 32 disparities, so numDisparities in (64, 256] is padded to the next multiple of 32 --
    // the reference's 218 to 224, not to 256 (round 5; before, nr was a multiple of 4: 128 / 256 / 384 / 512) -- and every
    // kernel moves and computes that much less.  MODE_SGBM_3WAY keeps whole groups of 8 disparities per lane (its
    // 8-slot tie rule is evaluated per lane), as does everything beyond 256.
    if (g->D > 64) {
        g->lanes = 16;
        g->nr = (g->D <= 256 && p->mode != CAMD_MODE_SGBM_3WAY) ? (g->D + 31) / 32 : 4 * ((g->D + 127) / 128);
    }
    else if (g->D > 32) { g->lanes = 8; g->nr = 4; }
    else if (g->D > 16) { g->lanes = 4; g->nr = 4; }
    else { g->lanes = 2; g->nr = 4; }
    g->Dp = g->lanes * 2 * g->nr;
    return CAMD_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Row ranges of the cost volume and the geometry the aggregation kernels see.  MODE_SGBM_3WAY: cv2's four fixed row
// stripes (oracle/sgbm_ref.c compute_disparity_3way), each with its warm-up overlap, each a "virtual pair".
static int cost_ranges(const Geom& g, int block_size_raw, CostRanges* cr, int* stripe_sz, int* max_rows)
{
    if (g.mode != CAMD_MODE_SGBM_3WAY) {
        cr->n = 1;
        cr->start[0] = 0; cr->rows[0] = g.H;
        for (int i = 1; i < 4; i++) { cr->start[i] = 0; cr->rows[i] = 0; }
        *stripe_sz = g.H;
        *max_rows = g.H;
        return CAMD_OK;
    }
    const int ns = 4, sz = div_up(g.H, ns), ov = (block_size_raw / 2 + 1) + div_up(sz, 10);
    cr->n = ns;
    *stripe_sz = sz;
    *max_rows = 0;
    for (int s = 0; s < ns; s++) {
        int a = s * sz - ov, b = (s + 1) * sz < g.H ? (s + 1) * sz : g.H;
        if (s > 0 && s * sz < g.H && a < 0) {
            set_error("image height %d too small for MODE_SGBM_3WAY with blockSize %d (a stripe of %d rows needs a warm-up "
                      "of %d rows above it)", g.H, block_size_raw, sz, ov);
            return CAMD_ERR_UNSUPPORTED;
        }
        a = a < 0 ? 0 : (a > g.H ? g.H : a);
        cr->start[s] = a;
        cr->rows[s] = b > a ? b - a : 0;
        if (cr->rows[s] > *max_rows) *max_rows = cr->rows[s];
    }
    return CAMD_OK;
}

// U7: an int16 overflow of the box sums is possible at all only beyond this bound
static bool params_may_overflow(const Geom& g)
{
    const long long K = 2 * g.SW2 + 1;
    return K * K * g.cn * (2 * g.ftzero + 63) + g.P2 > 32767;
}

// work of one pair in units of one 1080p / D=128 volume
static double pair_work(const Geom& g) { return ((double)g.H * g.W1 * g.Dp) / (1080.0 * 1792.0 * 128.0); }
static double auto_concurrent_limit(const Geom& g) { return g.mode == CAMD_MODE_HH ? 8.0 : 4.0; }
// The band passes win on throughput once their workgroups -- bands x (virtual) pairs -- fill the chip; below that
// the wavefront of a pass fills and drains over a mostly idle GPU and the line scans win (concurrently into their own
// volumes while there is room for them, one launch per direction otherwise).  Measured crossovers (r03_batch_sweep,
// tools/gpu_mid_d_paths.sh, tools/gpu_small_d_paths.sh, tools/gpu_get_depth_sweep.py): 1080p D=128 between 3 and 4
// pairs (117 / 156 workgroups), MODE_HH at 8 (312); 720p D=128 8 pairs (208) band +29 %; 720p D=64 8 pairs (104)
// scans +19 %; 4K D=64 2 pairs (78) scans.  The two-wavefront modes need about twice as many.
static int band_fill_workgroups(const Geom& g) { return (g.mode == CAMD_MODE_HH || g.mode == CAMD_MODE_HH4) ? 300 : 150; }
// largest batch the AUTO rule sends down the concurrent-direction path (it needs npaths volumes per pair): the
// batches whose band passes would not fill the chip; at least one pair, at most CAMD_MULTI_MAX_BATCH
static int auto_concurrent_pairs(const Geom& g, bool band_ok, int max_batch, int nbands)
{
    int cap = max_batch < CAMD_MULTI_MAX_BATCH ? max_batch : CAMD_MULTI_MAX_BATCH;
    int n;
    if (band_ok && nbands > 0) n = (band_fill_workgroups(g) - 1) / nbands;  // pairs with fewer workgroups than that
    else n = (int)(auto_concurrent_limit(g) / pair_work(g));                   // no band instantiation (D > 256)
    if (n < 1) n = 1;
    return n < cap ? n : cap;
}

// The band passes are instantiated for 16 lanes x {1,2} vectors and 8 / 4 / 2 lanes x 1 vector per pixel (every
// numDisparities up to 256); their inline winner-take-all (not used by MODE_SGBM_3WAY, which decides its winners in
// k_wta) covers uniquenessRatio <= 99.
static bool band_supported(const Geom& g)
{
    const bool shape = g.W1 > 0 && g.nr <= 8;  // (lanes < 16 always come with nr == 4)
    return shape && (g.mode == CAMD_MODE_SGBM_3WAY || g.uniq <= 99);
}

// ndirs directions in one launch (ndirs > 1 only with FIRST: each direction writes its own volume)
// One kernel instantiation per line-group shape (lanes, nr): M(LANES, NR) for the handle's shape.  16 lanes take
// nr = 3 .. 8 (numDisparities up to 256 in steps of 32), 12 and 16 (up to 384 / 512, scan kernels only).
#define CAMD_FOR_SHAPE(g, M)                      \
    do {                                          \
        if ((g).lanes == 2) M(2, 4);              \
        else if ((g).lanes == 4) M(4, 4);         \
        else if ((g).lanes == 8) M(8, 4);         \
        else switch ((g).nr) {                    \
            case 3: M(16, 3); break;              \
            case 4: M(16, 4); break;              \
            case 5: M(16, 5); break;              \
            case 6: M(16, 6); break;              \
            case 7: M(16, 7); break;              \
            case 8: M(16, 8); break;              \
            case 12: M(16, 12); break;            \
            default: M(16, 16);                   \
        }                                         \
    } while (0)
// the same for the band passes (nr <= 8; band_supported)
#define CAMD_FOR_BAND_SHAPE(g, M)                 \
    do {                                          \
        if ((g).lanes == 2) M(2, 4);              \
        else if ((g).lanes == 4) M(4, 4);         \
        else if ((g).lanes == 8) M(8, 4);         \
        else switch ((g).nr) {                    \
            case 3: M(16, 3); break;              \
            case 4: M(16, 4); break;              \
            case 5: M(16, 5); break;              \
            case 6: M(16, 6); break;              \
            case 7: M(16, 7); break;              \
            default: M(16, 8);                    \
        }                                         \
    } while (0)

template <bool FIRST>
static int launch_scan(const camd_sgbm* h, const int (*dirs)[2], int ndirs, uint16_t* S, size_t dir_stride,
                       int batch, hipStream_t st)
{
    const Geom& g = h->ga;
    ScanDirs sd;
    int maxlines = 0;
    for (int i = 0; i < 8; i++) {
        int k = i < ndirs ? i : 0;
        sd.dx[i] = dirs[k][0];
        sd.dy[i] = dirs[k][1];
        sd.nlines[i] = sd.dy[i] == 0 ? g.H : (sd.dx[i] == 0 ? g.W1 : g.W1 + g.H - 1);
        if (i < ndirs && sd.nlines[i] > maxlines) maxlines = sd.nlines[i];
    }
    sd.dir_stride = dir_stride;
    dim3 grid(div_up((long long)maxlines * g.lanes, 256), batch, ndirs);
    const bool pad = g.Dp != g.D;
#define CAMD_SCAN(LN, NVV)                                                                                  \
    do {                                                                                                    \
        if (pad) hipLaunchKernelGGL((k_scan<LN, NVV, FIRST, true>), grid, dim3(256), 0, st, h->C, S, g, sd, \
                                    h->vol_elems);                                                          \
        else hipLaunchKernelGGL((k_scan<LN, NVV, FIRST, false>), grid, dim3(256), 0, st, h->C, S, g, sd,    \
                                h->vol_elems);                                                              \
    } while (0)
    CAMD_FOR_SHAPE(g, CAMD_SCAN);
#undef CAMD_SCAN
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

static int launch_wta(const camd_sgbm* h, const uint16_t* S, int nvol, size_t dir_stride, int16_t* disp,
                      size_t pitch_e, size_t stride_e, int batch, hipStream_t st)
{
    const Geom& g = h->ga;
    const int tie_lanes = g.mode == CAMD_MODE_SGBM_3WAY ? h->way3_simd_lanes : 0;
    dim3 grid(g.H, batch);
    size_t lds = (size_t)g.W * 6;
    lds = align_up(lds, 16);
#define CAMD_WTA(LN, NVV)                                                                       \
    hipLaunchKernelGGL((k_wta<LN, NVV>), grid, dim3(256), lds, st, S, disp, pitch_e, stride_e, g, \
                       h->vol_elems, nvol, dir_stride, tie_lanes)
    CAMD_FOR_SHAPE(g, CAMD_WTA);
#undef CAMD_WTA
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

// one band-wavefront pass over `batch` (virtual) pairs (sgbm_band.hpp): full = H, V, Dg, A of sweep (sx, sy);
// !full = the row-parallel H-only pass.  mode 0 writes S, 1 adds to S, 2 reads S and decides the winners.
static int launch_band(camd_sgbm* h, int sx, int sy, bool full, int mode, int batch, hipStream_t st, bool diag = true,
                       bool tie8 = false)
{
    const Geom& g = h->ga;
    BandArgs a;
    a.C = h->C; a.S = h->S; a.E = h->E; a.flags = h->flags; a.ticket = h->ticket; a.err = h->err;
    a.keys = h->keys; a.d1 = h->d1; a.vol_stride = h->vol_elems; a.erec_stride = h->erec_stride;
    a.sx = sx; a.sy = sy; a.nbands = h->nbands; a.nchunks = h->nchunks; a.npairs = batch;
    a.epoch = ++h->epoch;
    a.write_S = h->keep_S;
    CAMD_HIP(hipMemsetAsync(h->ticket, 0, 4, st));
    // full passes: one workgroup per (pair, band); the row-parallel pass: the batch's rows in runs of R (sgbm_band.hpp)
    dim3 grid(full ? h->nbands * batch : div_up((long long)batch * g.H, (CAMD_BAND_ROW_ALL_WAVES ? BAND_BLOCK : BAND_THREADS) / g.lanes)), block(BAND_BLOCK);
    const bool pad = g.Dp != g.D;
    // (the shapes that exist since round 5, nr = 3 / 5 / 6 / 7, are instantiated in their padded form only: the
    // unpadded one merely skips two masking operations per register, and D = Dp is the rare case there)
#define CAMD_BAND(LN, NRR, FF, MM, DG)                                                                          \
    do {                                                                                                        \
        if (pad || (NRR) % 4 != 0) hipLaunchKernelGGL((k_band<LN, NRR, FF, MM, true, DG>), grid, block, 0, st, a, g); \
        else hipLaunchKernelGGL((k_band<LN, ((NRR) % 4 ? 4 : (NRR)), FF, MM, false, DG>), grid, block, 0, st, a, g);   \
    } while (0)
#define CAMD_BAND_F0T(LN, NRR) CAMD_BAND(LN, NRR, true, 0, true)
#define CAMD_BAND_F2T(LN, NRR) CAMD_BAND(LN, NRR, true, 2, true)
#define CAMD_BAND_F0F(LN, NRR) CAMD_BAND(LN, NRR, true, 0, false)
#define CAMD_BAND_F2F(LN, NRR) CAMD_BAND(LN, NRR, true, 2, false)
#define CAMD_BAND_R2(LN, NRR) CAMD_BAND(LN, NRR, false, 2, true)
#define CAMD_BAND_R1(LN, NRR) CAMD_BAND(LN, NRR, false, 1, true)
    if (full && mode == 0 && diag) CAMD_FOR_BAND_SHAPE(g, CAMD_BAND_F0T);
    else if (full && mode == 2 && diag) CAMD_FOR_BAND_SHAPE(g, CAMD_BAND_F2T);
    else if (full && mode == 0) CAMD_FOR_BAND_SHAPE(g, CAMD_BAND_F0F);
    else if (full && mode == 2) CAMD_FOR_BAND_SHAPE(g, CAMD_BAND_F2F);
    else if (!full && mode == 2 && tie8) {
        // MODE_SGBM_3WAY: nr is a multiple of 4 (normalise)
#define CAMD_BAND_TIE(LN, NRR)                                                                                  \
    do {                                                                                                        \
        if (pad) hipLaunchKernelGGL((k_band<LN, NRR, false, 2, true, true, true>), grid, block, 0, st, a, g);   \
        else hipLaunchKernelGGL((k_band<LN, NRR, false, 2, false, true, true>), grid, block, 0, st, a, g);      \
    } while (0)
        if (g.lanes == 16 && g.nr == 4) CAMD_BAND_TIE(16, 4);
        else if (g.lanes == 16) CAMD_BAND_TIE(16, 8);
        else if (g.lanes == 8) CAMD_BAND_TIE(8, 4);
        else if (g.lanes == 4) CAMD_BAND_TIE(4, 4);
        else CAMD_BAND_TIE(2, 4);
#undef CAMD_BAND_TIE
    }
    else if (!full && mode == 2 && h->persist_row > 0 && g.lanes == 16 && g.nr == 4) {
        // CAMD_OPT_RESIDENT: a fixed number of resident workgroups, runs of rows by ticket (sgbm_band.hpp)
        const dim3 pgrid(h->persist_row * (h->num_cus > 0 ? h->num_cus : 256));
        if (pad) hipLaunchKernelGGL((k_band_row_persist<16, 4, true>), pgrid, block, 0, st, a, g);
        else hipLaunchKernelGGL((k_band_row_persist<16, 4, false>), pgrid, block, 0, st, a, g);
    }
    else if (!full && mode == 2) CAMD_FOR_BAND_SHAPE(g, CAMD_BAND_R2);
    else if (!full && mode == 1) CAMD_FOR_BAND_SHAPE(g, CAMD_BAND_R1);
    else { set_error("band pass (full %d, mode %d) not instantiated", (int)full, mode); return CAMD_ERR_UNSUPPORTED; }
#undef CAMD_BAND_F0T
#undef CAMD_BAND_F2T
#undef CAMD_BAND_F0F
#undef CAMD_BAND_F2F
#undef CAMD_BAND_R2
#undef CAMD_BAND_R1
#undef CAMD_BAND
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

// A flagged volume for which no exact workspace could be allocated: never hand back a silently different result --
// its disparities are written as invalid and the handle reports an error (err = 2).
__global__ __launch_bounds__(256) void k_poison_flagged(int16_t* __restrict__ raw, size_t stride_e, size_t n,
                                                        const uint32_t* __restrict__ neg, uint32_t* __restrict__ err,
                                                        int invalid)
{
    const int vp = blockIdx.y;
    if (!neg[vp]) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        raw[(size_t)vp * stride_e + i] = (int16_t)invalid;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, 2u);
}

// the aggregation + winner-take-all in int arithmetic (sgbm_exact.hpp) of every FLAGGED (virtual) pair of the batch, in
// one launch that returns at once when nothing is flagged.  dst: the raw disparity image of (virtual) pair 0.
static int launch_exact(const camd_sgbm* h, int nvolumes, int16_t* dst, size_t stride_e, size_t n_e, hipStream_t st)
{
    const Geom& g = h->ga;
    // directions in the order OpenCV adds them up (it matters once sums saturate with negative terms in play)
    static const int d_sgbm[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {1, -1}, {0, -1}, {-1, -1}};
    static const int d_hh4[4][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}};   // v, ^, ->, <-  (oracle/sgbm_ref.c:532-607)
    static const int d_3way[3][2] = {{-1, 0}, {1, 0}, {0, 1}};           // (right + left) + top  (:789)
    const int (*dirs)[2] = g.mode == CAMD_MODE_HH4 ? d_hh4 : (g.mode == CAMD_MODE_SGBM_3WAY ? d_3way : d_sgbm);
    const int nd = g.npaths;
    ScanDirs sd;
    for (int i = 0; i < 8; i++) {
        const int k = i < nd ? i : 0;
        sd.dx[i] = dirs[k][0];
        sd.dy[i] = dirs[k][1];
        sd.nlines[i] = sd.dy[i] == 0 ? g.H : (sd.dx[i] == 0 ? g.W1 : g.W1 + g.H - 1);
    }
    sd.dir_stride = h->vol_elems;  // (int elements)
    const bool way3 = g.mode == CAMD_MODE_SGBM_3WAY;
    ExactArgs a;
    a.C = reinterpret_cast<const int16_t*>(h->C);
    a.Lx = h->Lx;
    a.dst = dst;
    a.vol_stride = h->vol_elems;
    a.dst_stride_e = stride_e;
    a.dst_pitch_e = (size_t)g.W;
    a.dst_n = n_e;
    a.neg = h->cost_neg;
    a.bar = h->xbar;
    a.err = h->err;
    a.nvol = nvolumes;
    a.nd = nd;
    a.tie_lanes = way3 ? h->way3_simd_lanes : 0;
    a.combine = g.mode == CAMD_MODE_HH4 ? 1 : (way3 ? 2 : 0);
    a.min_as_int = g.mode == CAMD_MODE_HH4 ? 1 : 0;
    a.invalid = (g.minD - 1) * 16;
    CAMD_HIP(hipMemsetAsync(h->xbar, 0, 4, st));
    const size_t lds = align_up((size_t)g.W * 6, 16);
    // One workgroup per compute unit and a hand-rolled grid barrier between the phases: the workgroups must all be
    // resident at once.  A cooperative launch makes the runtime check that (it refuses a grid that cannot be); where it is
    // refused or unsupported the plain launch runs with the barrier's bounded wait as the safety net (error bit 2).
    const dim3 grid(h->num_cus > 0 ? h->num_cus : 256);
    Geom gg = g;
    void* kargs[] = {(void*)&a, (void*)&gg, (void*)&sd};
#define CAMD_XALL(LN, NRR)                                                                                          \
    do {                                                                                                            \
        const void* fn = way3 ? (const void*)k_exact_all<LN, NRR, true> : (const void*)k_exact_all<LN, NRR, false>; \
        if (hipLaunchCooperativeKernel(fn, grid, dim3(256), kargs, lds, st) != hipSuccess) {                        \
            (void)hipGetLastError();                                                                                \
            if (way3) hipLaunchKernelGGL((k_exact_all<LN, NRR, true>), grid, dim3(256), lds, st, a, g, sd);         \
            else hipLaunchKernelGGL((k_exact_all<LN, NRR, false>), grid, dim3(256), lds, st, a, g, sd);             \
        }                                                                                                           \
    } while (0)
    CAMD_FOR_SHAPE(g, CAMD_XALL);
#undef CAMD_XALL
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

}  // namespace camd

using namespace camd;

extern "C" {

size_t camd_sgbm_workspace_bytes(const camd_sgbm_params* p, int width, int height, int channels,
                                 int max_batch)
{
    Geom g;
    if (normalise(p, width, height, channels, &g) != CAMD_OK || max_batch <= 0) return 0;
    size_t w1 = g.W1 > 0 ? (size_t)g.W1 : 0;
    CostRanges cr;
    int stripe_sz, vrows;
    if (cost_ranges(g, p->blockSize, &cr, &stripe_sz, &vrows) != CAMD_OK) return 0;
    const bool way3 = g.mode == CAMD_MODE_SGBM_3WAY;
    size_t vol = align_up((size_t)vrows * w1 * g.Dp * 2, 256);
    size_t raw = align_up((size_t)height * width * 2, 256);
    size_t total = (size_t)max_batch * (2 * cr.n * vol + raw);
    total += (size_t)max_batch * cr.n * 8 + 8;  // per-volume flags (near-overflow, below-P2), ticket + error word
    if (way3) total += (size_t)max_batch * cr.n * align_up((size_t)vrows * width * 2, 256);
    if (g.speckleWindowSize > 0) total += speckle_ws_bytes(width, height, max_batch);
    const bool band_ok = band_supported(g);
    if (band_ok) {
        const int R = BAND_THREADS / g.lanes;
        size_t nb = (size_t)div_up(vrows, R);
        total += (size_t)max_batch * cr.n * nb * (band_erec_stride(g.W1, g.lanes, (g.nr + 3) / 4) * 8 + (size_t)div_up(g.W1, BAND_CHUNK) * 4);
        total += (size_t)max_batch * cr.n * vrows * width * 6;
    }
    // per-direction volumes: the latency path's, and one set of int volumes for the exact aggregation where the
    // parameters allow an int16 overflow of the cost volume (sgbm_exact.hpp)
    if (!way3)
        total += (size_t)g.npaths * auto_concurrent_pairs(g, band_ok, max_batch, div_up(vrows, BAND_THREADS / g.lanes)) * vol;
    if (w1 > 0 && params_may_overflow(g)) total += (size_t)g.npaths * vol * 2;
    return total;
}

int camd_sgbm_create(const camd_sgbm_params* p, int width, int height, int channels, int max_batch,
                     camd_sgbm** out)
{
    if (!out) { set_error("out is NULL"); return CAMD_ERR_BAD_ARG; }
    *out = nullptr;
    Geom g;
    int rc = normalise(p, width, height, channels, &g);
    if (rc != CAMD_OK) return rc;
    if (max_batch <= 0) { set_error("max_batch must be > 0"); return CAMD_ERR_BAD_ARG; }
    rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    camd_sgbm* h = new (std::nothrow) camd_sgbm();
    if (!h) { set_error("out of host memory"); return CAMD_ERR_NOMEM; }
    memset(h, 0, sizeof(*h));
    h->g = g;
    h->params = *p;
    h->max_batch = max_batch;
    h->way3_simd_lanes = 8;
    int vrows = height;
    rc = cost_ranges(g, p->blockSize, &h->cr, &h->stripe_sz, &vrows);
    if (rc != CAMD_OK) { delete h; return rc; }
    const bool way3 = g.mode == CAMD_MODE_SGBM_3WAY;
    if (way3 && 2 * g.SW2 + 1 > 11) {
        set_error("MODE_SGBM_3WAY is implemented for blockSize <= 11");
        delete h;
        return CAMD_ERR_UNSUPPORTED;
    }
    h->ga = g;
    h->ga.H = vrows;
    size_t w1 = g.W1 > 0 ? (size_t)g.W1 : 0;
    h->vol_elems = align_up((size_t)vrows * w1 * g.Dp * 2, 256) / 2;
    size_t raw_e = align_up((size_t)height * width * 2, 256) / 2;
    const size_t nvol = (size_t)max_batch * h->cr.n;  // volumes: one per (virtual) pair
    hipError_t e = hipSuccess;
    if (w1 > 0) {
        if (e == hipSuccess) e = hipMalloc((void**)&h->C, nvol * h->vol_elems * 2);
        // the padded disparities d >= D of C hold P2 from here on: k_cost's all-padding waves do not write (sgbm_cost.hpp)
        // (complete before the handle is handed out: the first compute may run on a stream the null stream does not order)
        // Only a padded layout has such waves.  The fill runs on a stream of its own and only that stream is waited
        // for: no device-wide synchronisation, other streams keep running.
        if (e == hipSuccess && g.Dp != g.D) {
            hipStream_t fs = nullptr;
            e = hipStreamCreateWithFlags(&fs, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipMemsetD16Async(h->C, (unsigned short)g.P2, nvol * h->vol_elems, fs);
            if (e == hipSuccess) e = hipStreamSynchronize(fs);
            if (fs) (void)hipStreamDestroy(fs);
        }
        // (S shifted against C by 256 B ... 1 MB so that the two streams of a pass do not touch the same offsets at the same
        // time: no difference, profiles/r06_s_offset.txt)
        if (e == hipSuccess) e = hipMalloc((void**)&h->S, nvol * h->vol_elems * 2);
        if (e == hipSuccess && way3) e = hipMalloc((void**)&h->rawv, nvol * align_up((size_t)vrows * width * 2, 256));
    }
    if (e == hipSuccess) e = hipMalloc((void**)&h->raw, (size_t)max_batch * raw_e * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&h->cost_ovf, nvol * 8);
    h->cost_neg = h->cost_ovf ? h->cost_ovf + nvol : nullptr;
    if (e == hipSuccess) e = hipMemset(h->cost_ovf, 0, nvol * 8);
    h->may_overflow = params_may_overflow(g);
    if (e == hipSuccess) e = hipMalloc((void**)&h->ticket, 16);
    if (e == hipSuccess) e = hipMemset(h->ticket, 0, 16);
    h->err = h->ticket ? h->ticket + 1 : nullptr;
    h->xbar = h->ticket ? h->ticket + 2 : nullptr;
    h->persist_cost = h->persist_row = 0;   // (ticket[3]: k_cost_persist's item counter)
    {
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            h->num_cus = ncu;
        else
            (void)hipGetLastError();
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->err_host, 4, hipHostMallocDefault);
    if (e == hipSuccess) *h->err_host = 0;
    size_t sws = speckle_ws_bytes(width, height, max_batch);
    if (e == hipSuccess && g.speckleWindowSize > 0) e = hipMalloc(&h->speckle_ws, sws);
    h->band_ok = band_supported(g);
    h->path = 0;
    h->saturate = 1;
    h->phases = 7;
    h->epoch = 0;
    if (h->band_ok) {
        const int R = BAND_THREADS / g.lanes;
        h->nbands = div_up(vrows, R);  // bands of one (virtual) pair
        h->nchunks = div_up(g.W1, BAND_CHUNK);
        h->erec_stride = band_erec_stride(g.W1, g.lanes, (g.nr + 3) / 4);
        size_t nflags = nvol * h->nbands * h->nchunks;
        size_t npix = nvol * vrows * width;  // the winner-take-all state of every (virtual) pair
        if (e == hipSuccess) e = hipMalloc((void**)&h->E, nvol * h->nbands * h->erec_stride * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&h->flags, nflags * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&h->keys, npix * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&h->d1, npix * 2);
        if (e == hipSuccess) e = hipMemset(h->flags, 0, nflags * 4);
    }
    // per-direction volumes, allocated last and allowed to fail: the latency path then has fewer pairs (or none: the
    // sequential scans take over), and without the one set the exact aggregation needs a flagged volume is refused
    // loudly (k_poison_flagged) instead
    if (e == hipSuccess && w1 > 0) {
        int cap = way3 ? 0 : auto_concurrent_pairs(g, h->band_ok, max_batch, h->nbands);
        while (cap > 0) {
            if (hipMalloc((void**)&h->Smulti, (size_t)g.npaths * cap * h->vol_elems * 2) == hipSuccess) break;
            (void)hipGetLastError();
            h->Smulti = nullptr;
            cap /= 2;
        }
        h->smulti_cap = cap;
        if (h->may_overflow) {
            if (hipMalloc((void**)&h->Lx, (size_t)g.npaths * h->vol_elems * 4) == hipSuccess) h->exact_cap = 1;
            else { (void)hipGetLastError(); h->Lx = nullptr; }
        }
    }
    if (e != hipSuccess) {
        set_error("workspace allocation failed: %s", hipGetErrorString(e));
        camd_sgbm_destroy(h);
        return e == hipErrorOutOfMemory ? CAMD_ERR_NOMEM : CAMD_ERR_HIP;
    }
    *out = h;
    return CAMD_OK;
}

// what the bits of the device-side error word mean (camd_sgbm_status / the next compute report them)
static const char* device_error_text(uint32_t e)
{
    static const char* kRefused =
        "a cost volume left the int16 regime of the aggregation kernels (values below P2 after an overflow of the box "
        "sums) and the handle has no workspace for the exact path; the disparities of that pair were written as invalid";
    static const char* kTimeout =
        "a band-wavefront pass timed out waiting for its upstream band; the disparities of that call were written as "
        "invalid";
    static const char* kBoth =
        "a band-wavefront pass timed out waiting for its upstream band (the disparities of that call were written as "
        "invalid) AND a cost volume left the int16 regime of the aggregation kernels with no workspace for the exact "
        "path (that pair was written as invalid)";
    static const char* kBarrier =
        "the exact int path's grid barrier timed out (its workgroups were not all resident at once -- e.g. on a stream "
        "restricted to few compute units while other kernels held them); the disparities of the flagged pairs were written "
        "as invalid";
    if (e & 4u) return kBarrier;
    return (e & 3u) == 3u ? kBoth : ((e & 2u) ? kRefused : kTimeout);
}

int camd_sgbm_destroy(camd_sgbm* h)
{
    if (!h) return CAMD_OK;
    if (h->ev_ok)
        for (int i = 0; i <= ST_COUNT; i++) (void)hipEventDestroy(h->ev[i]);
    (void)hipFree(h->C); (void)hipFree(h->S);
    (void)hipFree(h->raw); (void)hipFree(h->rawv); (void)hipFree(h->speckle_ws); (void)hipFree(h->cost_ovf);
    (void)hipFree(h->E); (void)hipFree(h->flags); (void)hipFree(h->ticket); (void)hipFree(h->keys);
    (void)hipFree(h->d1); (void)hipFree(h->Smulti); (void)hipFree(h->Lx);
    if (h->err_host) (void)hipHostFree(h->err_host);
    delete h;
    return CAMD_OK;
}

int camd_sgbm_query(const camd_sgbm* h, int* width1, int* D, int* Dp, int* minX1)
{
    if (!h) { set_error("handle is NULL"); return CAMD_ERR_BAD_ARG; }
    if (width1) *width1 = h->g.W1;
    if (D) *D = h->g.D;
    if (Dp) *Dp = h->g.Dp;
    if (minX1) *minX1 = h->g.minX1;
    return CAMD_OK;
}

int camd_sgbm_set_option(camd_sgbm* h, int option, int value)
{
    if (!h) { set_error("handle is NULL"); return CAMD_ERR_BAD_ARG; }
    if (option == CAMD_OPT_PATH && value >= CAMD_PATH_AUTO && value <= CAMD_PATH_CONCURRENT) {
        if (value == CAMD_PATH_CONCURRENT && h->g.W1 > 0) {
            // an explicit request may exceed what AUTO would use: grow the per-direction volumes here (an init-time
            // call), never inside the stream-ordered compute
            const int want = h->max_batch < CAMD_MULTI_MAX_BATCH ? h->max_batch : CAMD_MULTI_MAX_BATCH;
            if (want > h->smulti_cap) {
                CAMD_HIP(hipDeviceSynchronize());
                (void)hipFree(h->Smulti);
                h->Smulti = nullptr;
                h->smulti_cap = 0;
                hipError_t e = hipMalloc((void**)&h->Smulti, (size_t)h->g.npaths * want * h->vol_elems * 2);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    set_error("no memory for %d x %d per-direction volumes", h->g.npaths, want);
                    return CAMD_ERR_NOMEM;
                }
                h->smulti_cap = want;
            }
        }
        h->path = value;
    }
    else if (option == CAMD_OPT_KEEP_S) h->keep_S = value != 0;
    else if (option == CAMD_OPT_COST && value >= CAMD_COST_AUTO && value <= CAMD_COST_SPLIT) h->cost_path = value;
    else if (option == CAMD_OPT_SATURATE) h->saturate = value != 0;
    else if (option == CAMD_OPT_3WAY_SIMD_LANES && (value == 1 || value == 8)) h->way3_simd_lanes = value;
    else if (option == CAMD_OPT_PHASES && value >= 1 && value <= 7) h->phases = value;
    else if (option == CAMD_OPT_RESIDENT && value >= 0 && (value >> 4) <= 4 && (value & 15) <= 4) {
        h->persist_cost = value >> 4;
        h->persist_row = value & 15;
    }
    else if (option == CAMD_OPT_EXACT) {
        if (value != 0 && !h->Lx && h->may_overflow) {
            // the workspace could not be had when the handle was made: try again rather than stay in refuse mode silently
            if (hipMalloc((void**)&h->Lx, (size_t)h->g.npaths * h->vol_elems * 4) != hipSuccess) {
                (void)hipGetLastError();
                h->Lx = nullptr;
                set_error("no memory for the exact path's %d per-direction int volumes; the handle keeps refusing flagged pairs",
                          h->g.npaths);
                return CAMD_ERR_NOMEM;
            }
        }
        h->exact_cap = (value != 0 && h->Lx) ? 1 : 0;
    }
    else { set_error("unknown option %d / value %d", option, value); return CAMD_ERR_BAD_ARG; }
    return CAMD_OK;
}

int camd_sgbm_status(camd_sgbm* h, void* stream)
{
    if (!h) { set_error("handle is NULL"); return CAMD_ERR_BAD_ARG; }
    if (!h->err) return CAMD_OK;
    uint32_t e = 0;
    CAMD_HIP(hipMemcpyAsync(&e, h->err, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CAMD_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (e) {
        CAMD_HIP(hipMemsetAsync(h->err, 0, 4, (hipStream_t)stream));
        if (h->err_host) *(volatile uint32_t*)h->err_host = 0;
        set_error("%s", device_error_text(e));
        return CAMD_ERR_HIP;
    }
    return CAMD_OK;
}

int camd_sgbm_set_profiling(camd_sgbm* h, int enable)
{
    if (!h) { set_error("handle is NULL"); return CAMD_ERR_BAD_ARG; }
    if (enable && !h->ev_ok) {
        for (int i = 0; i <= ST_COUNT; i++) CAMD_HIP(hipEventCreate(&h->ev[i]));
        h->ev_ok = true;
    }
    h->profiling = enable != 0;
    return CAMD_OK;
}
int camd_sgbm_num_stages(void) { return ST_COUNT; }
const char* camd_sgbm_stage_name(int i) { return i >= 0 && i < ST_COUNT ? kStageNames[i] : ""; }
int camd_sgbm_get_profile(camd_sgbm* h, float* ms, int n)
{
    if (!h || !ms) { set_error("NULL argument"); return CAMD_ERR_BAD_ARG; }
    if (!h->ev_ok || !h->profiling) { set_error("profiling not enabled"); return CAMD_ERR_BAD_ARG; }
    CAMD_HIP(hipEventSynchronize(h->ev[ST_COUNT]));
    for (int i = 0; i < n && i < ST_COUNT; i++) CAMD_HIP(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    return CAMD_OK;
}

int camd_sgbm_compute(camd_sgbm* h, const uint8_t* left, const uint8_t* right, size_t pitch,
                      size_t image_stride, int16_t* disp, size_t disp_pitch, size_t disp_stride,
                      int batch, void* stream)
{
    if (!h || !left || !right || !disp) { set_error("NULL argument"); return CAMD_ERR_BAD_ARG; }
    const Geom& g = h->g;
    if (batch <= 0 || batch > h->max_batch) {
        set_error("batch %d outside [1, max_batch=%d]", batch, h->max_batch);
        return CAMD_ERR_BAD_ARG;
    }
    if (pitch < (size_t)g.W * g.cn || disp_pitch < (size_t)g.W * 2 || (disp_pitch & 1) || (disp_stride & 1)) {
        set_error("pitch too small or odd disparity pitch");
        return CAMD_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    // A device-side bounded wait that expired in an EARLIER call (its result was written as all-invalid by
    // k_lrcheck) is reported here without a synchronisation: the flag travels through a pinned mirror that an
    // async copy refreshes at the end of every band-path compute.
    if (h->err_host && *(volatile uint32_t*)h->err_host) {
        const uint32_t e = *(volatile uint32_t*)h->err_host;
        *(volatile uint32_t*)h->err_host = 0;
        CAMD_HIP(hipMemsetAsync(h->err, 0, 4, st));
        set_error("in an earlier compute on this handle: %s", device_error_text(e));
        return CAMD_ERR_HIP;
    }
    const size_t dpe = disp_pitch / 2, dse = disp_stride / 2;
    const bool prof = h->profiling && h->ev_ok;
#define MARK(i) do { if (prof) CAMD_HIP(hipEventRecord(h->ev[i], st)); } while (0)
    h->last_batch = batch;
    const size_t raw_stride = align_up((size_t)g.H * g.W * 2, 256) / 2;

    if (g.W1 <= 0) {
        // minX1 >= maxX1: everything is INVALID_DISP_SCALED; the median of a constant is the constant
        hipLaunchKernelGGL(k_fill_s16, dim3(div_up(g.W, 256), g.H, batch), dim3(256), 0, st, disp, dpe, dse,
                           g.W, g.H, (g.minD - 1) * 16);
        CAMD_LAUNCH_CHECK();
        return CAMD_OK;
    }

    // ---- matching cost volume C ------------------------------------------------------------------------------
    const int K = 2 * g.SW2 + 1;
    // U7: int16 overflow is possible at all only beyond this bound; below it SAT == wrap
    const bool may_overflow = h->may_overflow;
    const bool sat = h->saturate && may_overflow;
    const bool way3 = g.mode == CAMD_MODE_SGBM_3WAY;
    const int vbatch = batch * h->cr.n;  // volumes this call fills (3WAY: four stripes per pair)
    const bool fused = K <= 11 && (h->cost_path != CAMD_COST_SPLIT || way3);
    const bool do_cost = (h->phases & 1) != 0;
    MARK(ST_COST);
    if (fused && do_cost) {
        // waves per workgroup: one per 8 disparities, at least 4 (the staging needs up to 3 waves of lanes), at most
        // 8 for RGB (three 8-wave workgroups share a CU at 74 VGPRs: 17.5 instead of 20.2 ms per 64 pairs; a 16-wave
        // workgroup would have a CU to itself) and 16 for gray (fewer registers, and the staging per cell halves)
        // 16 disparities per lane (CAMD_COST_DL16): a workgroup of 8 waves then covers 128 disparities, so the BT operands
        // of a strip are staged once instead of once per 64-disparity block, at K x 8 ring registers per lane
        const bool dl16 = CAMD_COST_DL16 && K <= 5 && g.Dp >= 64 && (g.cn == 3 || CAMD_COST_DL16 > 1);
        const int dl = dl16 ? 16 : COST_DL;
        const int maxw = g.cn == 3 || dl16 ? CAMD_COST_MAX_WAVES_RGB : CAMD_COST_MAX_WAVES_GRAY;
        const int nw = g.Dp / dl < 4 ? 4 : (g.Dp / dl < maxw ? g.Dp / dl : maxw);
        const int ndblk = div_up(g.Dp, nw * dl);                        // disparity blocks of <= 128
        const int nstrips = div_up(g.W1, 64 - (K - 1));
        // row chunks: enough workgroups for ~32 rounds over the chip.  The saturating recurrence must start at row 0
        // (one chunk: few workgroups, a long walk) -- but it only differs from the wrapping one on images that drive a
        // window sum to within one horizontal sum of 32767, so where the true sums cannot wrap 16 bits the chunked
        // wrapping kernel runs first and flags the volumes that need the sequential kernel (sgbm_cost.hpp)
        const int tbound = K * g.cn * (2 * g.ftzero + 63);
        const bool two_stage = sat && (long long)K * tbound + g.P2 <= 65535;
        auto launch_cost = [&](bool sat_kernel, uint32_t* ovf, int thresh) {
            int nchunks = 1;
            if (!sat_kernel) {
                const long long per_chunk = (long long)nstrips * ndblk * vbatch;
                nchunks = div_up(8192, per_chunk);
                const int maxc = h->ga.H / 32 > 1 ? h->ga.H / 32 : 1;
                nchunks = nchunks < 1 ? 1 : (nchunks > maxc ? maxc : nchunks);
                // Few workgroups (one or two pairs per call): the launch is a handful of rounds over the chip's places
                // (256 CUs x 3 eight-wave / 1 sixteen-wave workgroups), so the LAST round's fill decides: take the chunk
                // count that minimises rounds x (rows walked per workgroup, incl. the K-1 rows every chunk recomputes).
                // One 1080p RGB pair: 33 chunks = 1980 workgroups = 2.6 rounds of 37 rows -> 25 chunks = 1500 = 2 of 48.
                const long long places = 256LL * (nw <= 8 ? 3 : 1);
                if (per_chunk * nchunks < 6 * places) {
                    long long best = -1;
                    for (int nc = 1; nc <= maxc; nc++) {
                        const long long cost = (long long)div_up(per_chunk * nc, places) * (div_up(h->ga.H, nc) + K - 1);
                        if (best < 0 || cost < best) { best = cost; nchunks = nc; }
                    }
                }
            }
            // CAMD_OPT_RESIDENT: a fixed number of resident workgroups take the same items by ticket (k_cost_persist), in
            // smaller row chunks so that the last items of the launch end together
            const bool resident = h->persist_cost > 0 && !sat_kernel && !ovf && h->cr.n == 1 && (K == 5 || K == 3) && !dl16 &&
                                  g.D % (nw * dl) == 0;
            if (resident) {
                const long long per_chunk = (long long)nstrips * ndblk * vbatch;
                const int maxc = h->ga.H / 32 > 1 ? h->ga.H / 32 : 1;
                nchunks = div_up(24LL * h->persist_cost * (h->num_cus > 0 ? h->num_cus : 256), per_chunk);
                nchunks = nchunks < 1 ? 1 : (nchunks > maxc ? maxc : nchunks);
            }
            const int rb = div_up(h->ga.H, nchunks);  // rows per chunk, in the longest range
            nchunks = div_up(h->ga.H, rb);
            dim3 grid(nstrips, nchunks * ndblk, vbatch), block(64 * nw);
            const size_t lds = cost_lds_bytes(g.cn, nw, dl);
            if (resident) {
                uint32_t* tk = h->ticket + 3;  // (zeroed below, before the launches)
                const int nx = nstrips, ny = nchunks * ndblk, nitems = nx * ny * vbatch;
                const dim3 pgrid(h->persist_cost * (h->num_cus > 0 ? h->num_cus : 256));
#define CAMD_COSTP(CNN, KK) hipLaunchKernelGGL((k_cost_persist<CNN, KK>), pgrid, block, lds, st, left, right, pitch, image_stride, \
                                               h->C, g, rb, nchunks, h->vol_elems, h->cr, tk, nx, ny, nitems)
                if (g.cn == 1) { if (K == 5) CAMD_COSTP(1, 5); else CAMD_COSTP(1, 3); }
                else { if (K == 5) CAMD_COSTP(3, 5); else CAMD_COSTP(3, 3); }
#undef CAMD_COSTP
                return;
            }
#define CAMD_COST_DLX(CNN, KK, SS, DLL)                                                                                   \
    hipLaunchKernelGGL((k_cost<CNN, KK, SS, DLL>), grid, block, lds, st, left, right, pitch, image_stride, h->C, g, rb, \
                       nchunks, h->vol_elems, h->cr, ovf, thresh, h->cost_neg)
#define CAMD_COST(CNN, KK, SS) CAMD_COST_DLX(CNN, KK, SS, COST_DL)
#define CAMD_COST16(CNN, KK, SS) do { if (dl16) CAMD_COST_DLX(CNN, KK, SS, 16); else CAMD_COST_DLX(CNN, KK, SS, COST_DL); } while (0)
#define CAMD_COST_K(CNN)                                                                         \
    switch (K) {                                                                                 \
        case 1: CAMD_COST16(CNN, 1, false); break;                                               \
        case 3: CAMD_COST16(CNN, 3, false); break;                                               \
        case 5: if (sat_kernel) CAMD_COST16(CNN, 5, true); else CAMD_COST16(CNN, 5, false); break; \
        case 7: if (sat_kernel) CAMD_COST(CNN, 7, true); else CAMD_COST(CNN, 7, false); break;   \
        case 9: if (sat_kernel) CAMD_COST(CNN, 9, true); else CAMD_COST(CNN, 9, false); break;   \
        default: if (sat_kernel) CAMD_COST(CNN, 11, true); else CAMD_COST(CNN, 11, false);       \
    }
            if (g.cn == 1) { CAMD_COST_K(1) } else { CAMD_COST_K(3) }
#undef CAMD_COST_K
#undef CAMD_COST16
#undef CAMD_COST
#undef CAMD_COST_DLX
        };
        if (may_overflow) CAMD_HIP(hipMemsetAsync(h->cost_neg, 0, (size_t)vbatch * 4, st));
        if (h->persist_cost > 0) CAMD_HIP(hipMemsetAsync(h->ticket + 3, 0, 4, st));
        if (two_stage) {
            CAMD_HIP(hipMemsetAsync(h->cost_ovf, 0, (size_t)vbatch * 4, st));
            launch_cost(false, h->cost_ovf, 32767 - tbound);
            CAMD_LAUNCH_CHECK();
            launch_cost(true, h->cost_ovf, -1);
        } else {
            launch_cost(sat, nullptr, -1);
        }
        CAMD_LAUNCH_CHECK();
    }
    MARK(ST_HSUM);
    if (!fused && do_cost) {
        int nseg = div_up(g.W1, HSUM_SEG), ndblk = div_up(g.Dp, 64);
        dim3 grid(nseg, g.H, batch), block(64 * ndblk);
        const int es = g.cn == 1 ? 4 : 12;
        const size_t maxr = HSUM_SEG + 2 * g.SW2 + ndblk * 64 + 2, maxl = HSUM_SEG + 2 * g.SW2 + 2;
        size_t lds = ((maxr + maxl) * es + (size_t)ndblk * (2 * g.SW2 + 1) * 64) * 4;
#define CAMD_HSUM(CNN, KK)                                                                                        \
    hipLaunchKernelGGL((k_hsum<CNN, KK>), grid, block, lds, st, left, right, pitch, image_stride, h->S, g, ndblk, \
                       h->vol_elems)
#define CAMD_HSUM_K(CNN)                                \
    switch (2 * g.SW2 + 1) {                            \
        case 1: CAMD_HSUM(CNN, 1); break;               \
        case 3: CAMD_HSUM(CNN, 3); break;               \
        case 5: CAMD_HSUM(CNN, 5); break;               \
        case 7: CAMD_HSUM(CNN, 7); break;               \
        case 9: CAMD_HSUM(CNN, 9); break;               \
        case 11: CAMD_HSUM(CNN, 11); break;             \
        default: CAMD_HSUM(CNN, 0);                     \
    }
        // measured: the register ring pays for gray (11.3 -> 9.5 ms per 64 pairs) but not for RGB, where the longer
        // unrolled body costs more than the two LDS operations it saves (20.2 -> 21.0 ms)
        if (g.cn == 1) { CAMD_HSUM_K(1) } else { CAMD_HSUM(3, 0); }
#undef CAMD_HSUM_K
#undef CAMD_HSUM
        CAMD_LAUNCH_CHECK();
    }

    MARK(ST_VSUM);
    if (!fused && do_cost) {
        size_t rowv = (size_t)g.W1 * (g.Dp / 8);
        const uint4* hs4 = reinterpret_cast<const uint4*>(h->S);
        uint4* c4 = reinterpret_cast<uint4*>(h->C);
        const size_t ring_lds = (size_t)K * 256 * sizeof(uint4);
        if (sat) {
            // the saturating recurrence runs from row 0 down the whole column (LDS-ring kernel, any K)
            const dim3 vgrid(div_up((long long)rowv, 256), 1, batch);
            hipLaunchKernelGGL(k_vsum<true>, vgrid, dim3(256), ring_lds, st, hs4, c4, g, h->vol_elems / 8, g.H);
        } else {
            const dim3 vgrid(div_up((long long)rowv, 256), div_up(g.H, VSUM_ROWS), batch);
            switch (K) {
#define CAMD_VSUM(KK) case KK: hipLaunchKernelGGL((k_vsum_reg<KK>), vgrid, dim3(256), 0, st, hs4, c4, g, h->vol_elems / 8); break
                CAMD_VSUM(1); CAMD_VSUM(3); CAMD_VSUM(5); CAMD_VSUM(7); CAMD_VSUM(9); CAMD_VSUM(11);
#undef CAMD_VSUM
                default:
                    hipLaunchKernelGGL(k_vsum<false>, vgrid, dim3(256), ring_lds, st, hs4, c4, g, h->vol_elems / 8, VSUM_ROWS);
            }
        }
        CAMD_LAUNCH_CHECK();
    }

    // Volumes that hold a value below P2 are outside the regime of the packed-u16 aggregation kernels (sgbm_exact.hpp).
    // The saturating cost kernel reports them itself; behind the wrapping kernels and the split pair one more pass over
    // C finds them (only where the parameters allow an overflow at all)
    if (do_cost && may_overflow && !(fused && sat)) {
        if (!fused) CAMD_HIP(hipMemsetAsync(h->cost_neg, 0, (size_t)vbatch * 4, st));
        hipLaunchKernelGGL(k_flag_below, dim3(512, 1, vbatch), dim3(256), 0, st, reinterpret_cast<const int16_t*>(h->C),
                           h->ga, h->vol_elems, h->cr, h->cost_neg);
        CAMD_LAUNCH_CHECK();
    }

    if (!(h->phases & 6)) {  // CAMD_OPT_PHASES: the caller runs the aggregation in a second call (another stream)
        for (int i = ST_SCAN; i <= ST_COUNT; i++) MARK(i);
        return CAMD_OK;
    }
    // (the split between the two aggregation passes exists on the plain band path only; elsewhere bits 1 and 2 travel together)
    const bool ph_first = (h->phases & 2) != 0, ph_last = (h->phases & 4) != 0;

    // aggregation path: fused band passes win on throughput (>= ~8 pairs per launch), concurrent
    // per-direction scans on latency (a few pairs: every direction gets its own S volume and all of them
    // run at once), sequential scans are the generic fallback
    // Measured at 1080p / D=128 (tools/gpu_batch_sweep.py, ms per pair): the concurrent scans win up to 4 pairs
    // per call for 5 paths (3.1 / 2.4 / 2.0 against 6.8 / 3.7 / 2.2 through the band passes) and up to 8 pairs
    // for 8 paths (3.6 ... 2.6 against 13.1 ... 2.9); from there on the band passes take over (1.36 / 1.87 at 16
    // pairs, 0.98 / 1.14 at 64).
    const int mcap = h->smulti_cap;
    int path = h->path;
    if (path == CAMD_PATH_AUTO) {
        // band passes when their workgroups fill the chip (band_fill_workgroups), else the concurrent scans
        if (h->band_ok && (long long)h->nbands * vbatch >= band_fill_workgroups(g)) path = CAMD_PATH_BAND;
        else path = CAMD_PATH_CONCURRENT;
    }
    if (way3) {
        // no per-direction volumes for the stripes: band passes, or three line scans + k_wta.  The wavefront of a
        // band pass fills slowly, so for little work the scans win (ms per pair, scans / band, 1080p D=128: 2.1 / 4.2
        // for one pair, 1.65 / 1.38 for four; VGA D=64: 0.36 / 1.09 for one, 0.125 / 0.10 for sixteen)
        if (h->path == CAMD_PATH_AUTO) path = batch * pair_work(g) < 2.5 ? CAMD_PATH_SCAN : CAMD_PATH_BAND;
        else if (path != CAMD_PATH_SCAN) path = CAMD_PATH_BAND;
    }
    if (path == CAMD_PATH_BAND && !h->band_ok) path = CAMD_PATH_SCAN;
    // the per-direction volumes were sized in create / set_option: a larger batch takes the next best path -- the band
    // passes when they at least come close to filling the chip, else one scan launch per direction
    if (path == CAMD_PATH_CONCURRENT && batch > mcap)
        path = (h->band_ok && (h->path != CAMD_PATH_AUTO || (long long)h->nbands * vbatch >= 128)) ? CAMD_PATH_BAND : CAMD_PATH_SCAN;
    // At 4 or 2 lanes per pixel (numDisparities <= 32) a band is 112 or 224 rows high: an image has only a handful of
    // bands (tools/gpu_small_d_paths.sh, scans / band passes in pairs/s: 1080p D=32 8 pairs 1585 / 1294, 16 pairs
    // 1673 / 2202; VGA D=16 16 pairs 9980 / 6430, 64 pairs 17060 / 18530): same rule, the band passes from ~128
    // workgroups on
    if (h->path == CAMD_PATH_AUTO && path == CAMD_PATH_BAND && g.lanes <= 4 && (long long)h->nbands * vbatch < 128)
        path = CAMD_PATH_SCAN;
    const bool band = path == CAMD_PATH_BAND;
    const bool multi = path == CAMD_PATH_CONCURRENT;
    const size_t dir_stride = (size_t)mcap * h->vol_elems;
    MARK(ST_SCAN);
    // 3WAY decides its winners inside the last band pass when that pass knows the tie rule in force: cv2's 8-slot
    // rule for D % 8 == 0, or the scalar build's "smallest d" (the ordinary rule); otherwise k_wta does it afterwards
    const bool way3_inline = way3 && (h->way3_simd_lanes == 1 || g.D % 8 == 0);
    if (band && way3) {
        // the stripes are independent "virtual pairs": -> and v in one band pass, <- by the row-parallel pass
        const Geom& ga = h->ga;
        if (way3_inline) {
            size_t npix = (size_t)vbatch * ga.H * ga.W;
            hipLaunchKernelGGL(k_wta_init, dim3(div_up((long long)npix, 256)), dim3(256), 0, st, h->keys, h->d1, npix,
                               (g.minD - 1) * 16);
            CAMD_LAUNCH_CHECK();
        }
        int rc = launch_band(h, +1, +1, true, 0, vbatch, st, false);
        if (rc != CAMD_OK) return rc;
        MARK(ST_SCAN2);
        rc = way3_inline ? launch_band(h, -1, +1, false, 2, vbatch, st, true, h->way3_simd_lanes == 8)
                         : launch_band(h, -1, +1, false, 1, vbatch, st);
        if (rc != CAMD_OK) return rc;
    } else if (band) {
        // fused passes: every pass reads C once and touches S once for up to four directions
        int rc = CAMD_OK;
        if (ph_first) {
            size_t npix = (size_t)batch * g.H * g.W;
            hipLaunchKernelGGL(k_wta_init, dim3(div_up((long long)npix, 256)), dim3(256), 0, st, h->keys, h->d1, npix,
                               (g.minD - 1) * 16);
            CAMD_LAUNCH_CHECK();
            rc = launch_band(h, +1, +1, true, 0, batch, st, g.mode != CAMD_MODE_HH4);          // ->  v  [\.  ./]
            if (rc != CAMD_OK) return rc;
        }
        MARK(ST_SCAN2);
        if (!ph_last) {
            for (int i = ST_WTA; i <= ST_COUNT; i++) MARK(i);
            return CAMD_OK;
        }
        if (g.mode == CAMD_MODE_HH4) rc = launch_band(h, -1, -1, true, 2, batch, st, false);    // <-  ^ + WTA
        else if (g.mode == CAMD_MODE_HH) rc = launch_band(h, -1, -1, true, 2, batch, st);       // <-  ^  \^  /^ + WTA
        else rc = launch_band(h, -1, +1, false, 2, batch, st);                                  // <- + WTA
        if (rc != CAMD_OK) return rc;
    } else {
        static const int dirs8[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {1, -1}, {0, -1}, {-1, -1}};
        static const int dirs4[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};  // MODE_SGBM_3WAY = the first three
        const int (*dirs)[2] = (g.mode == CAMD_MODE_HH4 || way3) ? dirs4 : dirs8;
        if (multi) {
            int rc = launch_scan<true>(h, dirs, g.npaths, h->Smulti, dir_stride, batch, st);
            if (rc != CAMD_OK) return rc;
        } else {
            for (int i = 0; i < g.npaths; i++) {
                int rc = i == 0 ? launch_scan<true>(h, dirs + i, 1, h->S, 0, vbatch, st)
                                : launch_scan<false>(h, dirs + i, 1, h->S, 0, vbatch, st);
                if (rc != CAMD_OK) return rc;
            }
        }
        MARK(ST_SCAN2);  // (no separate last pass on the scan paths: zero-length stage)
    }

    // Volumes flagged as outside the u16 regime are aggregated again in int arithmetic, one at a time through the one
    // set of per-direction volumes (two launches per volume that return at once for the others), and their raw
    // disparities replaced; without that workspace they are written as invalid and reported.
    auto exact_redo = [&](int16_t* dst, size_t stride_e, size_t n_e, int nvolumes) -> int {
        if (!may_overflow) return CAMD_OK;
        if (!h->exact_cap) {
            hipLaunchKernelGGL(k_poison_flagged, dim3(64, nvolumes), dim3(256), 0, st, dst, stride_e, n_e, h->cost_neg,
                               h->err, (g.minD - 1) * 16);
            CAMD_LAUNCH_CHECK();
            CAMD_HIP(hipMemcpyAsync(h->err_host, h->err, 4, hipMemcpyDeviceToHost, st));
            return CAMD_OK;
        }
        return launch_exact(h, nvolumes, dst, stride_e, n_e, st);
    };

    MARK(ST_WTA);
    if (band && !way3) {
        hipLaunchKernelGGL(k_lrcheck, dim3(div_up(div_up(g.W, 2), 256), div_up(g.H, LRCHECK_ROWS), batch), dim3(256), 0, st, h->d1, h->keys, h->raw,
                           (size_t)g.W, raw_stride, g, h->err);
        CAMD_LAUNCH_CHECK();
        CAMD_HIP(hipMemcpyAsync(h->err_host, h->err, 4, hipMemcpyDeviceToHost, st));
    } else if (way3) {
        // winner-take-all + LR check per stripe row, then every image row is taken from the stripe that owns it
        const size_t rawv_stride = align_up((size_t)h->ga.H * g.W * 2, 256) / 2;
        if (band && way3_inline) {
            hipLaunchKernelGGL(k_lrcheck, dim3(div_up(div_up(g.W, 2), 256), div_up(h->ga.H, LRCHECK_ROWS), vbatch), dim3(256), 0, st, h->d1, h->keys,
                               h->rawv, (size_t)g.W, rawv_stride, h->ga, h->err);
            CAMD_LAUNCH_CHECK();
        } else {
            int rc = launch_wta(h, h->S, 1, 0, h->rawv, (size_t)g.W, rawv_stride, vbatch, st);
            if (rc != CAMD_OK) return rc;
        }
        {
            int rc = exact_redo(h->rawv, rawv_stride, (size_t)h->ga.H * g.W, vbatch);
            if (rc != CAMD_OK) return rc;
        }
        hipLaunchKernelGGL(k_gather_stripes, dim3(div_up(g.W, 256), g.H, batch), dim3(256), 0, st, h->rawv, rawv_stride,
                           h->raw, raw_stride, g.W, g.H, h->stripe_sz, h->cr, band ? h->err : nullptr,
                           (may_overflow && !h->exact_cap) ? h->cost_neg : nullptr, (g.minD - 1) * 16);
        CAMD_LAUNCH_CHECK();
        if (band) CAMD_HIP(hipMemcpyAsync(h->err_host, h->err, 4, hipMemcpyDeviceToHost, st));
    } else {
        int rc = multi ? launch_wta(h, h->Smulti, g.npaths, dir_stride, h->raw, (size_t)g.W, raw_stride, batch, st)
                       : launch_wta(h, h->S, 1, 0, h->raw, (size_t)g.W, raw_stride, batch, st);
        if (rc != CAMD_OK) return rc;
    }

    if (!way3) {
        int rc = exact_redo(h->raw, raw_stride, (size_t)g.H * g.W, batch);
        if (rc != CAMD_OK) return rc;
    }

    MARK(ST_POST);
    {
        int rc = launch_median3(h->raw, g.W, raw_stride, disp, dpe, dse, g.W, g.H, batch, st);
        if (rc != CAMD_OK) return rc;
        MARK(ST_SPECKLE);
        if (g.speckleWindowSize > 0) {
            // the "workspace is clean" invariant is established by the previous call's k_cc_apply, in stream order: a call
            // on another stream is not ordered behind it, so it clears the workspace itself (on its own stream)
            if (h->speckle_stream != st) { h->speckle_clean = false; h->speckle_stream = st; }
            rc = launch_speckle(disp, dpe, dse, g.W, g.H, (g.minD - 1) * 16, g.speckleWindowSize,
                                16 * g.speckleRange, h->speckle_ws, speckle_ws_bytes(g.W, g.H, h->max_batch), batch, st,
                                &h->speckle_clean);
            if (rc != CAMD_OK) return rc;
        }
    }
    MARK(ST_COUNT);
#undef MARK
    return CAMD_OK;
}

int camd_sgbm_debug_copy(camd_sgbm* h, int which, int index, void* dst, void* stream)
{
    if (!h || !dst) { set_error("NULL argument"); return CAMD_ERR_BAD_ARG; }
    if (index < 0 || index >= h->max_batch) { set_error("index out of range"); return CAMD_ERR_BAD_ARG; }
    const Geom& g = h->g;
    hipStream_t st = (hipStream_t)stream;
    if (which == 0 || which == 1) {
        if (g.mode == CAMD_MODE_SGBM_3WAY) {
            set_error("MODE_SGBM_3WAY keeps its volumes per stripe: only which = 2 (raw disparity) is available");
            return CAMD_ERR_UNSUPPORTED;
        }
        if (g.W1 <= 0) return CAMD_OK;
        const uint16_t* src = (which == 0 ? h->C : h->S) + (size_t)index * h->vol_elems;
        CAMD_HIP(hipMemcpyAsync(dst, src, (size_t)g.H * g.W1 * g.Dp * 2, hipMemcpyDeviceToDevice, st));
    } else if (which == 2) {
        const size_t raw_stride = align_up((size_t)g.H * g.W * 2, 256) / 2;
        CAMD_HIP(hipMemcpyAsync(dst, h->raw + (size_t)index * raw_stride, (size_t)g.H * g.W * 2,
                                hipMemcpyDeviceToDevice, st));
    } else {
        set_error("which must be 0 (C), 1 (S) or 2 (raw disparity)");
        return CAMD_ERR_BAD_ARG;
    }
    return CAMD_OK;
}

}  // extern "C"

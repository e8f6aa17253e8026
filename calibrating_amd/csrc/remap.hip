// remap.hip -- cv2.remap / cv2.undistort on 8-bit images for gfx950.
//
// Replaces (file:line in /root/reference/calibrating/):
//   stereo_camera.py:217-228  cv2.remap(img, mapx, mapy, cv2.INTER_LANCZOS4)       (rectify, x2)
//   stereo_camera.py:230-240  the x-translation of rectify_img2 (x_shift, fused)
//   stereo_camera.py:430-431  cv2.undistort(img1, K, D)   (bilinear through 1/32-px fixed maps)
// Arithmetic follows OpenCV's 8-bit fixed-point remap: 1/32-pixel phases, 15-bit int16 weights from
// a 32x32-entry table (sum forced to 32768), int32 accumulate, (sum + 16384) >> 15, BORDER_CONSTANT 0.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

namespace camd {

enum { INTER_BITS = 5, INTER_TAB_SIZE = 32, COEF_BITS = 15, COEF_SCALE = 1 << 15 };

static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

// ---- fixed-point interpolation tables (host, built once per process and option value) ------------------------
// The table of a KS-tap kernel has 32 x 32 entries (one per 1/32-pixel phase pair), each KS*KS int16 weights
// = round(wy[ky] * wx[kx] * 32768) whose sum is then forced to exactly 32768.
//
// U15: which taps take that correction is a process-wide option shared with the oracle's switch
// of the same name, so the two cannot drift apart: the correction looks at the 2x2 taps (k1, k2) in
// [lo, lo+2) x [lo, lo+2) of the entry, adds the missing weight to the largest of them or removes the excess from
// the smallest.  lo = KS/2 by default (OpenCV's loop bounds as recalled; 4 for Lanczos-4), settable to KS/2 - 1
// through camd_set_global_option(CAMD_GOPT_LANCZOS_FIX_GROUP_LO, 3).
static int g_fix_group_lo8 = 4;

// Phase table of the 1-D kernels, [32][KS] float32.
static std::vector<float> phase_weights(int ks)
{
    std::vector<float> w((size_t)INTER_TAB_SIZE * ks);
    if (ks == 2) {  // bilinear: (1 - t, t)
        for (int p = 0; p < INTER_TAB_SIZE; p++) {
            const float t = p * (1.f / INTER_TAB_SIZE);
            w[p * 2] = 1.f - t;
            w[p * 2 + 1] = t;
        }
        return w;
    }
    // Lanczos-4, taps at offsets -3..4 from the integer position: w_k ~ sin(pi d) sin(pi d / 4) / d^2 with
    // d = t + 3 - k, evaluated the way OpenCV does -- sin(pi d) sin(pi d/4) is expanded into the sine and
    // cosine of ONE angle a = -(t+3) pi/4 with a table of eighth-turn rotations, in double, divided by (pi d / 4)^2,
    // rounded to float; a tap the sample falls on (|d| < 1e-6) gets 1e30; the eight floats are then scaled by the
    // float reciprocal of their float sum.
    const double q = 0.70710678118654752440084436210485, pi4 = 3.1415926535897932384626433832795 * 0.25;
    const double rot[8][2] = {{1, 0}, {-q, -q}, {0, 1}, {q, -q}, {-1, 0}, {q, q}, {0, -1}, {-q, q}};
    for (int p = 0; p < INTER_TAB_SIZE; p++) {
        const float t = p * (1.f / INTER_TAB_SIZE);
        const double a = -(t + 3) * pi4, sa = std::sin(a), ca = std::cos(a);
        float* o = &w[(size_t)p * 8];
        float total = 0;
        for (int k = 0; k < 8; k++) {
            const float d = t + 3 - k;
            if (std::fabs(d) < 1e-6f) o[k] = 1e30f;
            else {
                const double y = -d * pi4;
                o[k] = (float)((rot[k][0] * sa + rot[k][1] * ca) / (y * y));
            }
            total += o[k];
        }
        const float inv = 1.f / total;
        for (int k = 0; k < 8; k++) o[k] *= inv;
    }
    return w;
}

// [32*32][ks*ks] int16.  The correction of an entry is applied right after the entry is quantised, while the
// entries behind it are still zero: for the 2x2 table the window [lo, lo+2)^2 = [1, 3)^2 reaches past the entry
// into that zero region (as in cv2's builder), where it can only ever fire at phase (0,0) -- the weight 32768
// saturates to 32767 and the missing 1 lands on tap (1,1).
static void build_itab(int ks, int16_t* itab)
{
    const std::vector<float> w = phase_weights(ks);
    const int n = ks * ks, lo = ks == 8 ? g_fix_group_lo8 : ks / 2;
    const long total = (long)INTER_TAB_SIZE * INTER_TAB_SIZE * n;
    std::fill(itab, itab + total, (int16_t)0);
    for (int py = 0; py < INTER_TAB_SIZE; py++)
        for (int px = 0; px < INTER_TAB_SIZE; px++) {
            int16_t* e = itab + ((long)py * INTER_TAB_SIZE + px) * n;
            int sum = 0;
            for (int ky = 0; ky < ks; ky++)
                for (int kx = 0; kx < ks; kx++) {
                    const float v = w[py * ks + ky] * w[px * ks + kx];
                    sum += e[ky * ks + kx] = sat_short((int)lrintf(v * COEF_SCALE));
                }
            const int excess = sum - COEF_SCALE;
            if (excess == 0) continue;
            // largest / smallest tap of the 2x2 group, first one wins on ties (scan order ky, kx)
            const long room = total - (e - itab);
            int big = lo * ks + lo, small = big;
            for (int ky = lo; ky < lo + 2; ky++)
                for (int kx = lo; kx < lo + 2; kx++) {
                    const int i = ky * ks + kx;
                    if (i >= room) continue;
                    if (e[i] < e[small]) small = i;
                    else if (e[i] > e[big]) big = i;
                }
            const int at = excess < 0 ? big : small;
            e[at] = (short)(e[at] - excess);
        }
}

// device copies of the tables, one set per device, created on first use
struct DevTables {
    int16_t* lanczos = nullptr;
    int16_t* bilinear = nullptr;
};
static std::mutex g_tab_mutex;
static DevTables g_tabs[64];

static int get_tables(const int16_t** lanczos, const int16_t** bilinear)
{
    int dev = 0;
    CAMD_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index %d out of range", dev); return CAMD_ERR_HIP; }
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    DevTables& t = g_tabs[dev];
    if (!t.lanczos) {
        std::vector<int16_t> hl(1024 * 64), hb(1024 * 4);
        build_itab(8, hl.data());
        build_itab(2, hb.data());
        int16_t *dl = nullptr, *db = nullptr;
        CAMD_HIP(hipMalloc((void**)&dl, hl.size() * 2));
        CAMD_HIP(hipMalloc((void**)&db, hb.size() * 2));
        CAMD_HIP(hipMemcpy(dl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice));
        CAMD_HIP(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        t.lanczos = dl;
        t.bilinear = db;
    }
    *lanczos = t.lanczos;
    *bilinear = t.bilinear;
    return CAMD_OK;
}

// One row of a KS-wide window that starts `shb` bytes into raw[0]: funnel-shift the dwords into place
// (v_alignbyte), expand byte pairs to int16 pairs with one v_perm each and feed v_dot2_i32_i16 with the table's
// weight pairs -- 2 MACs per op instead of a byte load + mad per tap.
// MASKED: only window bytes [lo, hi) lie inside the image row; the rest become the constant border 0.
template <int KS, int CN, bool MASKED = false>
__device__ __forceinline__ void mac_window_row(const uint32_t (&raw)[(KS * CN + 3) / 4 + 1], uint32_t shb,
                                               const int16_t* __restrict__ wrow, int (&acc)[CN], int lo = 0,
                                               int hi = KS * CN)
{
    constexpr int NW = (KS * CN + 3) / 4;
    uint32_t win[NW + 1];
#pragma unroll
    for (int j = 0; j < NW; j++) {
        win[j] = __builtin_amdgcn_alignbyte(raw[j + 1], raw[j], shb);
        if (MASKED) {
            const int nlo = min(max(lo - 4 * j, 0), 4), nhi = min(max(hi - 4 * j, 0), 4);
            win[j] &= (uint32_t)((1ull << (8 * nhi)) - 1) & ~(uint32_t)((1ull << (8 * nlo)) - 1);
        }
    }
    win[NW] = 0;
    const uint32_t* wr = reinterpret_cast<const uint32_t*>(wrow);  // KS/2 weight pairs
#pragma unroll
    for (int q = 0; q < KS / 2; q++) {
        const uint32_t wq = wr[q];
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const int b0 = (2 * q) * CN + c, j0 = b0 / 4, o0 = b0 % 4, o1 = o0 + CN;  // compile-time after unrolling
            const uint32_t sel = 0x0c000c00u | (uint32_t)o0 | ((uint32_t)o1 << 16);
            const uint32_t pr = __builtin_amdgcn_perm(win[j0 + 1], win[j0], sel);    // (tap 2q | tap 2q+1 << 16)
            acc[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, pr), __builtin_bit_cast(s16x2_t, wq), acc[c],
                                            false);
        }
    }
}

template <int CN>
__device__ __forceinline__ void store_rounded(const int (&acc)[CN], uint8_t* out)
{
#ifdef CAMD_REMAP_DBG_NOSTORE  // measurement only (tools/microtests/remap_bench.hip): keep the arithmetic, drop the stores
    if (acc[0] != 0x12345678) return;
#endif
#pragma unroll
    for (int c = 0; c < CN; c++) {
        int v = (acc[c] + (1 << (COEF_BITS - 1))) >> COEF_BITS;
        out[c] = (uint8_t)min(max(v, 0), 255);
    }
}

// One row of a KS-wide window whose first byte is win[0]'s byte 0 (the interior path fetches the row from its own
// byte address, so nothing has to be shifted into place); weights come from registers.
template <int KS, int CN>
__device__ __forceinline__ void mac_row_regs(const uint32_t (&win)[(KS * CN + 3) / 4 + 1], const uint32_t* wq,
                                             int (&acc)[CN])
{
#pragma unroll
    for (int q = 0; q < KS / 2; q++) {
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const int b0 = (2 * q) * CN + c, j0 = b0 / 4, o0 = b0 % 4, o1 = o0 + CN;  // compile-time after unrolling
            const uint32_t sel = 0x0c000c00u | (uint32_t)o0 | ((uint32_t)o1 << 16);
            const uint32_t pr = __builtin_amdgcn_perm(win[j0 + 1], win[j0], sel);    // (tap 2q | tap 2q+1 << 16)
            acc[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, pr), __builtin_bit_cast(s16x2_t, wq[q]),
                                            acc[c], false);
        }
    }
}

// One destination pixel of `nz` images that share a map (a batch of one rig): KS x KS taps at (ix, iy) ..
// (ix+KS-1, iy+KS-1).  Everything that depends only on the map -- cell, phase, the weight entry (in registers:
// wreg), the interior test, the byte offset of the window -- is worked out once and applied to every image.
// A window row is fetched as aligned dwords (x4 + x3) and funnel-shifted into place; HR rows are requested at a time
// (KS / 2 for Lanczos: with the entry in registers and half the rows in flight the kernel needs 80 VGPRs and no LDS
// while it walks the images, so six waves per SIMD cover the gather's latency).
// Measured on 64 1080p RGB images, 16 images per workgroup (tools/microtests/remap_bench.hip; one image per workgroup
// and the entry behind an LDS slab, round 2's form: 2.95 ms): entry in LDS / registers at 117 VGPRs and 4 waves per
// SIMD 2.06 ms; registers, all 8 rows at once (97 VGPRs) 1.63; 4 rows at a time, 6 waves per SIMD 1.48; 2 rows at a
// time, 7 waves 1.68.  Slower and dropped: byte-exact misaligned row fetches 3.17, prefetching the next image's rows
// 2.56 (7 ms when capped at 128 VGPRs), staging the source box in LDS through registers 2.16 or by global_load_lds 3.5.
template <int KS, int CN, int HR>
__device__ __forceinline__ void gather_pixel_batch(const uint8_t* __restrict__ src, int sw, int sh, size_t pitch,
                                                   size_t src_stride, int ix, int iy,
                                                   const uint32_t (&wreg)[KS * KS / 2], uint8_t* out, size_t dst_stride,
                                                   int nz)
{
    constexpr int NB = KS * CN, NW = (NB + 3) / 4, NL = NW + 1;
    constexpr int OVER = (4 * (NW + 1) - NB + CN - 1) / CN, UNDER = (3 + CN - 1) / CN;
    const bool interior = ix >= UNDER && iy >= 0 && ix + KS + OVER <= sw && iy + KS <= sh;
    if (__all(interior)) {
        const uint8_t* pw = src + (size_t)iy * pitch + (size_t)ix * CN;
        const uint32_t shb = (uint32_t)(reinterpret_cast<uintptr_t>(pw) & 3);
        const uint8_t* p0 = pw - shb;
        uint32_t raw[HR][NL];
#pragma unroll 1
        for (int z = 0; z < nz; z++, p0 += src_stride, out += dst_stride) {
            int acc[CN];
#pragma unroll
            for (int c = 0; c < CN; c++) acc[c] = 0;
#pragma unroll
            for (int r0 = 0; r0 < KS; r0 += HR) {
#pragma unroll
                for (int r = 0; r < HR; r++) __builtin_memcpy(raw[r], p0 + (size_t)(r0 + r) * pitch, 4 * NL);
#pragma unroll
                for (int r = 0; r < HR; r++) {
                    uint32_t win[NW + 1];
#pragma unroll
                    for (int j = 0; j < NW; j++) win[j] = __builtin_amdgcn_alignbyte(raw[r][j + 1], raw[r][j], shb);
                    win[NW] = 0;
                    mac_row_regs<KS, CN>(win, wreg + (r0 + r) * (KS / 2), acc);
                }
                if (HR < KS) __builtin_amdgcn_sched_barrier(0);  // keep the next portion's loads behind this portion's use
            }
            store_rounded<CN>(acc, out);
        }
        return;
    }
    const bool touches = !(ix >= sw || ix + KS <= 0 || iy >= sh || iy + KS <= 0);
    const int lo = max(0, -ix * CN), hi = min(KS * CN, (sw - ix) * CN);
#pragma unroll 1
    for (int z = 0; z < nz; z++, src += src_stride, out += dst_stride) {
        int acc[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) acc[c] = 0;
        if (touches) {
            const uintptr_t img_lo = reinterpret_cast<uintptr_t>(src);
            const uintptr_t img_hi = img_lo + (size_t)(sh - 1) * pitch + (size_t)sw * CN;
#pragma unroll
            for (int r = 0; r < KS; r++) {  // (unrolled: the weights are registers, their index must be static)
                const int yy = iy + r;
                if (yy < 0 || yy >= sh) continue;
                const uintptr_t pa = img_lo + (uintptr_t)((long long)yy * (long long)pitch + (long long)ix * CN);
                const uintptr_t b = pa & ~(uintptr_t)3;
                uint32_t rw[NW + 1];
                if (b >= img_lo && b + 4 * (NW + 1) <= img_hi) {
#pragma unroll
                    for (int j = 0; j <= NW; j++) rw[j] = reinterpret_cast<const uint32_t*>(b)[j];
                } else {
#pragma unroll
                    for (int j = 0; j <= NW; j++) {
                        uint32_t v = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uintptr_t q = b + 4 * j + k;
                            if (q >= img_lo && q < img_hi) v |= (uint32_t)*reinterpret_cast<const uint8_t*>(q) << (8 * k);
                        }
                        rw[j] = v;
                    }
                }
                mac_window_row<KS, CN, true>(rw, (uint32_t)(pa & 3), reinterpret_cast<const int16_t*>(wreg + r * (KS / 2)), acc,
                                             lo, hi);
            }
        }
        store_rounded<CN>(acc, out);
    }
}

// cv2.remap with CV_32FC1 maps; blockIdx.z owns `zb` consecutive images of the batch.  What bounded the Lanczos
// case first was fetching each pixel's own 128-byte weight entry: eight 16-byte loads per lane, every one touching
// 64 different cache lines.  So a wave fetches its 64 entries cooperatively into a wave-private LDS slab and every
// lane reads its own entry back -- in two halves of 64 bytes (four lanes per half entry: 64 contiguous bytes; slab
// stride 80 B: conflict-free ds_read_b128), 5 KB of LDS per wave, none of it needed once the entry is in registers.
#ifndef CAMD_REMAP_ROWS_PER_FETCH
#define CAMD_REMAP_ROWS_PER_FETCH 4
#endif
template <int KS, int CN, int HR = (KS == 8 ? CAMD_REMAP_ROWS_PER_FETCH : KS)>
__global__ __launch_bounds__(256, KS != 8 ? 1 : HR == 8 ? 4 : HR == 4 ? 6 : 7)
void k_remap_f32(const uint8_t* __restrict__ src, int sw, int sh, size_t src_pitch, size_t src_stride,
                 const float* __restrict__ mapx, const float* __restrict__ mapy, uint8_t* __restrict__ dst, int dw,
                 int dh, size_t dst_pitch, size_t dst_stride, const int16_t* __restrict__ tab, int x_shift, int batch,
                 int zb)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, xm = x - x_shift;
    const int z0 = blockIdx.z * zb, nz = min(zb, batch - z0);
    const bool act = x < dw && xm >= 0 && xm < dw;
    int a = 0, ix = 0, iy = 0;
    if (act) {
        const size_t mi = (size_t)y * dw + xm;
        // RemapInvoker: float map * 32 in float, cvRound (half to even), split into cell and phase
        const int sx = __float2int_rn(mapx[mi] * (float)INTER_TAB_SIZE);
        const int sy = __float2int_rn(mapy[mi] * (float)INTER_TAB_SIZE);
        a = (sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1));
        ix = min(max(sx >> INTER_BITS, -32768), 32767) - (KS / 2 - 1);
        iy = min(max(sy >> INTER_BITS, -32768), 32767) - (KS / 2 - 1);
    }
    uint32_t wreg[KS * KS / 2];
    if constexpr (KS == 8) {
        constexpr int HS = 80;
        __shared__ __attribute__((aligned(16))) uint8_t s_h[4][64 * HS];
        const int lane = threadIdx.x & 63;
        uint8_t* slab = s_h[threadIdx.x >> 6];
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = i * 16 + (lane >> 2);  // the lane whose half entry this group of four lanes fetches
                const int ae = __shfl(a, e);
                const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)ae * 64 + half * 32 + (lane & 3) * 8);
                *reinterpret_cast<uint4*>(slab + e * HS + (lane & 3) * 16) = v;
            }
            // the slab is private to this wave and LDS executes a wave's operations in order: only the compiler
            // has to be kept from moving the reads above the writes (and the next half's writes above the reads)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = *reinterpret_cast<const uint4*>(slab + lane * HS + q * 16);
                wreg[half * 16 + 4 * q] = v.x;
                wreg[half * 16 + 4 * q + 1] = v.y;
                wreg[half * 16 + 4 * q + 2] = v.z;
                wreg[half * 16 + 4 * q + 3] = v.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
        for (int i = 0; i < KS * KS / 2; i++) wreg[i] = reinterpret_cast<const uint32_t*>(tab + (size_t)a * (KS * KS))[i];
    }
    if (x >= dw) return;
    uint8_t* out = dst + (size_t)z0 * dst_stride + (size_t)y * dst_pitch + (size_t)x * CN;
    if (!act) {  // the columns the x-shift leaves empty
        for (int z = 0; z < nz; z++, out += dst_stride) {
#pragma unroll
            for (int c = 0; c < CN; c++) out[c] = 0;
        }
        return;
    }
    gather_pixel_batch<KS, CN, HR>(src + (size_t)z0 * src_stride, sw, sh, src_pitch, src_stride, ix, iy, wreg, out,
                                   dst_stride, nz);
}

template <int CN>
__global__ __launch_bounds__(256) void k_remap_nearest_u8(const uint8_t* __restrict__ src, int sw, int sh,
                                                          size_t src_pitch, size_t src_stride,
                                                          const float* __restrict__ mapx,
                                                          const float* __restrict__ mapy,
                                                          uint8_t* __restrict__ dst, int dw, int dh,
                                                          size_t dst_pitch, size_t dst_stride, int x_shift)
{
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y;
    if (x >= dw) return;
    uint8_t* out = dst + (size_t)blockIdx.z * dst_stride + (size_t)y * dst_pitch + (size_t)x * CN;
    int xm = x - x_shift;
    bool ok = xm >= 0 && xm < dw;
    int sx = 0, sy = 0;
    if (ok) {
        size_t mi = (size_t)y * dw + xm;
        sx = min(max(__float2int_rn(mapx[mi]), -32768), 32767);
        sy = min(max(__float2int_rn(mapy[mi]), -32768), 32767);
        ok = (unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh;
    }
    const uint8_t* p = src + (size_t)blockIdx.z * src_stride + (size_t)sy * src_pitch + (size_t)sx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) out[c] = ok ? p[c] : 0;
}

template <int CN>
__global__ __launch_bounds__(256) void k_remap_fixed_bilinear(const uint8_t* __restrict__ src, int sw, int sh,
                                                              size_t src_pitch, size_t src_stride,
                                                              const int16_t* __restrict__ mapxy,
                                                              const uint16_t* __restrict__ mapa,
                                                              uint8_t* __restrict__ dst, int dw, int dh,
                                                              size_t dst_pitch, size_t dst_stride,
                                                              const int16_t* __restrict__ tab, int batch, int zb)
{
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y;
    if (x >= dw) return;
    const int z0 = blockIdx.z * zb, nz = min(zb, batch - z0);
    size_t mi = (size_t)y * dw + x;
    int ix = mapxy[mi * 2], iy = mapxy[mi * 2 + 1];
    int a = mapa[mi] & (INTER_TAB_SIZE * INTER_TAB_SIZE - 1);
    const uint32_t* e = reinterpret_cast<const uint32_t*>(tab + (size_t)a * 4);
    const uint32_t wreg[2] = {e[0], e[1]};
    gather_pixel_batch<2, CN, 2>(src + (size_t)z0 * src_stride, sw, sh, src_pitch, src_stride, ix, iy, wreg,
                                 dst + (size_t)z0 * dst_stride + (size_t)y * dst_pitch + (size_t)x * CN, dst_stride, nz);
}

// images per workgroup of the batch-inner kernels: all of them (up to 16) while the grid still fills the chip
static int images_per_group(int groups_per_image, int batch)
{
    int zb = std::min(batch, 16);
    while (zb > 1 && (long long)groups_per_image * div_up(batch, zb) < 4096) zb = (zb + 1) / 2;
    return zb;
}

// host: one row of initUndistortRectifyMap's inner loop (X/Y/W accumulate per column, float64)
struct Dist { double k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4; };

static void inv3(const double m[9], double o[9])
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

}  // namespace camd

using namespace camd;

extern "C" {

int camd_set_global_option(int option, int value)
{
    if (option == CAMD_GOPT_LANCZOS_FIX_GROUP_LO && (value == 3 || value == 4)) {
        std::lock_guard<std::mutex> lock(g_tab_mutex);
        if (value != g_fix_group_lo8) {
            g_fix_group_lo8 = value;
            for (DevTables& t : g_tabs) {  // rebuilt on next use (the old copies may still be read by running kernels)
                t.lanczos = nullptr;
                t.bilinear = nullptr;
            }
        }
        return CAMD_OK;
    }
    set_error("unknown global option %d / value %d", option, value);
    return CAMD_ERR_BAD_ARG;
}

int camd_lanczos4_table_host(int16_t* tab)
{
    if (!tab) { set_error("NULL table"); return CAMD_ERR_BAD_ARG; }
    build_itab(8, tab);
    return CAMD_OK;
}

int camd_bilinear_table_host(int16_t* tab)
{
    if (!tab) { set_error("NULL table"); return CAMD_ERR_BAD_ARG; }
    build_itab(2, tab);
    return CAMD_OK;
}

int camd_remap_u8(const uint8_t* src, int sw, int sh, int cn, size_t src_pitch, size_t src_stride,
                  const float* mapx, const float* mapy, uint8_t* dst, int dw, int dh, size_t dst_pitch,
                  size_t dst_stride, int interp, int x_shift, int batch, void* stream)
{
    if (!src || !mapx || !mapy || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0 ||
        (cn != 1 && cn != 3) || src_pitch < (size_t)sw * cn || dst_pitch < (size_t)dw * cn) {
        set_error("camd_remap_u8: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    const int16_t *tl = nullptr, *tb = nullptr;
    rc = get_tables(&tl, &tb);
    if (rc != CAMD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(div_up(dw, 256), dh, batch), block(256);
    const int zb = images_per_group(grid.x * grid.y, batch);
    const dim3 gridz(grid.x, grid.y, div_up(batch, zb));
#define ARGS src, sw, sh, src_pitch, src_stride, mapx, mapy, dst, dw, dh, dst_pitch, dst_stride
    if (interp == CAMD_INTER_LANCZOS4) {
        if (cn == 1) hipLaunchKernelGGL((k_remap_f32<8, 1>), gridz, block, 0, st, ARGS, tl, x_shift, batch, zb);
        else hipLaunchKernelGGL((k_remap_f32<8, 3>), gridz, block, 0, st, ARGS, tl, x_shift, batch, zb);
    } else if (interp == CAMD_INTER_LINEAR) {
        if (cn == 1) hipLaunchKernelGGL((k_remap_f32<2, 1>), gridz, block, 0, st, ARGS, tb, x_shift, batch, zb);
        else hipLaunchKernelGGL((k_remap_f32<2, 3>), gridz, block, 0, st, ARGS, tb, x_shift, batch, zb);
    } else if (interp == CAMD_INTER_NEAREST) {
        if (cn == 1) hipLaunchKernelGGL((k_remap_nearest_u8<1>), grid, block, 0, st, ARGS, x_shift);
        else hipLaunchKernelGGL((k_remap_nearest_u8<3>), grid, block, 0, st, ARGS, x_shift);
    } else {
        set_error("camd_remap_u8: interpolation %d not implemented", interp);
        return CAMD_ERR_UNSUPPORTED;
    }
#undef ARGS
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_remap_fixed_bilinear_u8(const uint8_t* src, int sw, int sh, int cn, size_t src_pitch,
                                 size_t src_stride, const int16_t* mapxy, const uint16_t* mapa, uint8_t* dst,
                                 int dw, int dh, size_t dst_pitch, size_t dst_stride, int batch, void* stream)
{
    if (!src || !mapxy || !mapa || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0 ||
        (cn != 1 && cn != 3) || src_pitch < (size_t)sw * cn || dst_pitch < (size_t)dw * cn) {
        set_error("camd_remap_fixed_bilinear_u8: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    const int16_t *tl = nullptr, *tb = nullptr;
    rc = get_tables(&tl, &tb);
    if (rc != CAMD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int zb = images_per_group(div_up(dw, 256) * dh, batch);
    dim3 grid(div_up(dw, 256), dh, div_up(batch, zb)), block(256);
    if (cn == 1)
        hipLaunchKernelGGL((k_remap_fixed_bilinear<1>), grid, block, 0, st, src, sw, sh, src_pitch, src_stride,
                           mapxy, mapa, dst, dw, dh, dst_pitch, dst_stride, tb, batch, zb);
    else
        hipLaunchKernelGGL((k_remap_fixed_bilinear<3>), grid, block, 0, st, src, sw, sh, src_pitch, src_stride,
                           mapxy, mapa, dst, dw, dh, dst_pitch, dst_stride, tb, batch, zb);
    CAMD_LAUNCH_CHECK();
    return CAMD_OK;
}

int camd_undistort_maps_host(const double K[9], const double* dist, int ndist, int w, int h,
                             int16_t* mapxy, uint16_t* mapa)
{
    if (!K || !mapxy || !mapa || w <= 0 || h <= 0 || ndist < 0 || ndist > 14 || (ndist > 0 && !dist)) {
        set_error("camd_undistort_maps_host: bad arguments");
        return CAMD_ERR_BAD_ARG;
    }
    double dv[14] = {0};
    for (int i = 0; i < ndist; i++) dv[i] = dist[i];
    if (ndist > 12 && (dv[12] != 0. || dv[13] != 0.)) {
        set_error("tilted-sensor distortion (tauX, tauY) not implemented");
        return CAMD_ERR_UNSUPPORTED;
    }
    const Dist k = {dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7], dv[8], dv[9], dv[10], dv[11]};
    // cv2.undistort works in stripes of rows and folds the stripe offset into the new camera matrix
    int stripe0 = (1 << 12) / (w > 1 ? w : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > h) stripe0 = h;
    double Ar[9], ir[9];
    for (int i = 0; i < 9; i++) Ar[i] = K[i];
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5], cy0 = K[5];
    for (int y = 0; y < h; y += stripe0) {
        int stripe = stripe0 < h - y ? stripe0 : h - y;
        Ar[5] = cy0 - y;
        inv3(Ar, ir);
        for (int i = 0; i < stripe; i++) {
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            int16_t* mxy = mapxy + (size_t)(y + i) * w * 2;
            uint16_t* ma = mapa + (size_t)(y + i) * w;
            for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
                double ww = 1. / _w, x = _x * ww, yy = _y * ww;
                double x2 = x * x, y2 = yy * yy;
                double r2 = x2 + y2, _2xy = 2 * x * yy;
                double kr = (1 + ((k.k3 * r2 + k.k2) * r2 + k.k1) * r2) /
                            (1 + ((k.k6 * r2 + k.k5) * r2 + k.k4) * r2);
                double xd = (x * kr + k.p1 * _2xy + k.p2 * (r2 + 2 * x2) + k.s1 * r2 + k.s2 * r2 * r2);
                double yd = (yy * kr + k.p1 * (r2 + 2 * y2) + k.p2 * _2xy + k.s3 * r2 + k.s4 * r2 * r2);
                double u = fx * xd + u0, v = fy * yd + v0;
                int iu = (int)lrint(u * INTER_TAB_SIZE), iv = (int)lrint(v * INTER_TAB_SIZE);
                mxy[j * 2] = (int16_t)(iu >> INTER_BITS);
                mxy[j * 2 + 1] = (int16_t)(iv >> INTER_BITS);
                ma[j] = (uint16_t)((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1)));
            }
        }
    }
    return CAMD_OK;
}

}  // extern "C"

// sgbm_exact.hpp -- the aggregation outside the packed-u16 regime, in plain 32-bit integer arithmetic.
// Included by sgbm.hip (shares Geom, ScanDirs, CostRanges).
//
// k_scan / k_band compute  L = C + min(Lp[d], Lp[d-1] + P1, Lp[d+1] + P1, delta) - delta  on packed unsigned 16-bit
// halves, which is exact while 0 <= L <= 32767, i.e. while every C(y, x, d) >= P2 (then L >= C - P2 >= 0).  A cost
// volume can leave that regime in one way only: after an int16 overflow of the box-sum recurrence that builds it
// (saturating like OpenCV's CV_SIMD build, or wrapping like the scalar build: CAMD_OPT_SATURATE) the recurrence keeps
// what it lost, C drifts below P2 or turns negative, and L goes negative with it.  OpenCV then simply carries on in int:
//     L (int) = C + min(...) - delta,  Lr[d] = (CostType)L,  minLr = (CostType) min_d L,
//     S = saturate_cast<CostType>(S + L0 + L1 + L2 + L3)                       (oracle/sgbm_ref.c:351-374, :396-408)
// so for the volumes a cost kernel (or k_flag_below) has flagged as holding a value below P2 the host re-runs the
// aggregation with the kernels below -- element-wise int arithmetic, one line per lane group like k_scan, every
// direction into its own volume, and a winner-take-all that adds the directions up in OpenCV's grouping.  Correctness
// over speed: this path only ever runs for adversarial inputs (tools/gpu_fuzz.py: 2 of 4000 random cases).
#pragma once

namespace camd {

// raises neg[volume] when any C(y, x, d < D) of the volume's own rows is below P2 (as int16).  Used where the cost
// kernel that built the volume does not track it itself (the wrapping kernels; the split pair).
__global__ __launch_bounds__(256) void k_flag_below(const int16_t* __restrict__ C, Geom g, size_t vol_stride,
                                                    CostRanges cr, uint32_t* __restrict__ neg)
{
    const int vp = blockIdx.z, rows = cr.rows[vp % cr.n];
    const size_t n8 = (size_t)rows * g.W1 * (g.Dp / 8);
    const uint4* p = reinterpret_cast<const uint4*>(C + (size_t)vp * vol_stride);
    const uint32_t thr = dup16((uint32_t)g.P2);
    bool below = false;
    const int dq = g.Dp / 8;  // uint4 per pixel
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        uint4 q = p[i];
        // padded d >= D are expected to hold P2, but nothing here depends on it: they are replaced by the threshold
        const int d0 = (int)(i % (size_t)dq) * 8;
        if (d0 + 8 > g.D) {
            uint32_t* w = &q.x;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int d = d0 + 2 * k;
                if (d >= g.D) w[k] = thr;
                else if (d + 1 >= g.D) w[k] = (w[k] & 0xffffu) | (thr & 0xffff0000u);
            }
        }
        const s16x2_t t = __builtin_bit_cast(s16x2_t, thr);
        const s16x2_t m = __builtin_elementwise_min(
            __builtin_elementwise_min(__builtin_bit_cast(s16x2_t, q.x), __builtin_bit_cast(s16x2_t, q.y)),
            __builtin_elementwise_min(__builtin_bit_cast(s16x2_t, q.z), __builtin_bit_cast(s16x2_t, q.w)));
        below |= m.x < t.x || m.y < t.y;
    }
    if (__any(below) && (threadIdx.x & 63) == 0) atomicOr(neg + vp, 1u);
}

template <int LANES>
__device__ __forceinline__ int group_min_i32(int v)
{
    if (LANES >= 2) v = min(v, (int)dpp_mov<DPP_QUAD_XOR1>((uint32_t)v, (uint32_t)v));
    if (LANES >= 4) v = min(v, (int)dpp_mov<DPP_QUAD_XOR2>((uint32_t)v, (uint32_t)v));
    if (LANES >= 8) v = min(v, (int)dpp_mov<DPP_ROW_HALF_MIRROR>((uint32_t)v, (uint32_t)v));
    if (LANES >= 16) v = min(v, (int)dpp_mov<DPP_ROW_MIRROR>((uint32_t)v, (uint32_t)v));
    return v;
}

__device__ __forceinline__ int sat16i(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// One direction `dir` of sd, ONE volume; L of that direction goes to Lout + dir * sd.dir_stride AS INT, before
// it is narrowed: OpenCV's sums S += L0 + L1 + L2 + L3 take the int values, not the stored CostType ones, and once C
// is below -32768 + P2 the two differ (wta_row<EXACT> narrows as the mode's own loop does).
// STORE_SAT: how L is narrowed to int16 for the recursion itself -- false: the (CostType) cast of
// computeDisparitySGBM / computeDisparitySGBM_HH4 (wraps), true: the saturate_cast of the 3-way loop
// (oracle/sgbm_ref.c:768-774), whose minimum is taken over the narrowed values.
// (`dir` = index into sd, `unit` = which run of 256 / LANES lines of that direction this workgroup scans)
template <int LANES, int NR, bool STORE_SAT>
__device__ __forceinline__ void scan_exact_unit(const int16_t* __restrict__ Cv, int32_t* __restrict__ Lout, const Geom& g,
                                                const ScanDirs& sd, int dir, int unit, int min_as_int)
{
    constexpr int NE = 2 * NR;  // disparities per lane
    const int dx = sd.dx[dir], dy = sd.dy[dir], nlines = sd.nlines[dir];
    int32_t* __restrict__ Lv = Lout + (size_t)dir * sd.dir_stride;
    const int tid = unit * 256 + threadIdx.x;
    const int line = tid / LANES, li = tid % LANES;
    if (line >= nlines) return;  // (whole groups: 256 % LANES == 0)
    const int W1 = g.W1, H = g.H;
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
            const int k = line - W1 + 1;
            y0 = dy > 0 ? k : H - 1 - k;
        }
    }
    const int lx = dx == 0 ? (1 << 30) : (dx > 0 ? W1 - x0 : x0 + 1);
    const int ly = dy == 0 ? (1 << 30) : (dy > 0 ? H - y0 : y0 + 1);
    const int len = min(lx, ly);
    const size_t off = ((size_t)y0 * W1 + x0) * g.Dp + (size_t)li * NE;
    const ptrdiff_t step = ((ptrdiff_t)dy * W1 + dx) * (ptrdiff_t)g.Dp;
    const int16_t* cp = Cv + off;
    int32_t* lp_out = Lv + off;
    const int dbase = li * NE;

    int Lp[NE];
#pragma unroll
    for (int j = 0; j < NE; j++) Lp[j] = 0;
    int minLp = 0;
    auto load8 = [&](const int16_t* p, int (&dst)[NE]) {
        uint32_t w[NR];
        ld_regs<NR>(reinterpret_cast<const uint16_t*>(p), w);
#pragma unroll
        for (int k = 0; k < NR; k++) {
            dst[2 * k] = (int)(int16_t)(w[k] & 0xffffu);
            dst[2 * k + 1] = (int)(int16_t)(w[k] >> 16);
        }
    };
    int c[NE], cn[NE];
    load8(cp, c);
    for (int i = 0; i < len; i++) {
        if (i + 1 < len) load8(cp + (ptrdiff_t)(i + 1) * step, cn);
        // d-1 of my first element / d+1 of my last element live in the neighbouring lanes of the group
        int left = (int)dpp_mov<DPP_ROW_SHR1>((uint32_t)MAX_COST, (uint32_t)Lp[NE - 1]);
        int right = (int)dpp_mov<DPP_ROW_SHL1>((uint32_t)MAX_COST, (uint32_t)Lp[0]);
        if (li == 0) left = MAX_COST;
        if (li == LANES - 1) right = MAX_COST;
        const int delta = minLp + g.P2;
        int L[NE], Lint[NE], mn = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int d = dbase + j;
            const int lm = d == 0 ? MAX_COST : (j == 0 ? left : Lp[j > 0 ? j - 1 : 0]);
            const int lq = d >= g.D - 1 ? MAX_COST : (j == NE - 1 ? right : Lp[j < NE - 1 ? j + 1 : j]);
            const int v = c[j] + min(min(lm + g.P1, lq + g.P1), min(Lp[j], delta)) - delta;
            const int st = STORE_SAT ? sat16i(v) : (int)(int16_t)v;
            L[j] = d < g.D ? st : 0;
            Lint[j] = d < g.D ? v : 0;
            if (d < g.D) mn = min(mn, STORE_SAT ? st : v);
        }
        mn = group_min_i32<LANES>(mn);
        // computeDisparitySGBM stores the minimum as CostType; computeDisparitySGBM_HH4 hands it on as the int it is
        // (oracle/sgbm_ref.c:370-374 against :478-488, :545)
        minLp = (STORE_SAT || min_as_int) ? mn : (int)(int16_t)mn;
        int2* o = reinterpret_cast<int2*>(lp_out + (ptrdiff_t)i * step);  // 8-byte aligned: li * 2NR ints
#pragma unroll
        for (int v = 0; v < NR; v++) o[v] = make_int2(Lint[2 * v], Lint[2 * v + 1]);
#pragma unroll
        for (int j = 0; j < NE; j++) { Lp[j] = L[j]; c[j] = cn[j]; }
    }
}

// ---- one launch for the whole batch ------------------------------------------------------------------------------
// Rounds 3-4 queued two launches per volume (scan, winner-take-all) that returned at once unless the volume was flagged:
// 128 empty launches per 64-pair batch of the reference's default matcher, whose parameters merely ALLOW an overflow
// (0.6 ms of a 27 ms batch).  Now ONE persistent kernel walks the flags: nothing flagged (the normal case) -> every
// workgroup reads nvol words and leaves.  A flagged volume is scanned by all workgroups together (units of 256 / LANES
// lines, grid-stride), a grid-wide barrier, its rows are decided (grid-stride), a second barrier (the one set of
// per-direction volumes is reused by the next flagged volume).  The grid is one workgroup per compute unit, so all of it
// is resident once whatever else runs on the device has made room; the barrier is a monotonic counter (zeroed by the
// host before the launch) with agent-scope fences on both sides, and every wait is bounded: on expiry the error word is
// raised (camd_sgbm_status / the next compute report it) and the flagged volumes are written as invalid.
struct ExactArgs {
    const int16_t* C;      // volume 0
    int32_t* Lx;           // npaths per-direction int volumes (one set)
    int16_t* dst;          // raw disparity image of volume 0
    size_t vol_stride;     // int16 elements between volumes of C
    size_t dst_stride_e;   // elements between the disparity images of consecutive volumes
    size_t dst_pitch_e;
    size_t dst_n;          // elements of one disparity image
    const uint32_t* neg;   // per-volume flags
    uint32_t* bar;         // grid barrier counter
    uint32_t* err;
    int nvol, nd, tie_lanes, combine, min_as_int, invalid;
};
static constexpr uint32_t EXACT_SPIN_LIMIT = 1u << 23;  // seconds of s_sleep polling: a flagged 4K volume takes ~0.1 s a phase

template <int LANES, int NR, bool STORE_SAT>
__global__ __launch_bounds__(256) void k_exact_all(ExactArgs a, Geom g, ScanDirs sd)
{
    uint32_t target = 0;
    bool dead = false;
    auto grid_sync = [&]() {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();  // release: this workgroup's stores reach the device-wide level before it is counted
            target += gridDim.x;
            atomicAdd(a.bar, 1u);
            uint32_t spins = 0;
            while (!dead && __hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > EXACT_SPIN_LIMIT) {
                    atomicOr(a.err, 4u);  // bit 2: this grid barrier (bit 0 is a band pass's upstream wait)
                    dead = true;
                }
            }
            __threadfence();  // acquire: the other workgroups' stores are visible from here on
        }
        __syncthreads();
    };
    int units[8], total = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        units[d] = d < a.nd ? (sd.nlines[d] * LANES + 255) / 256 : 0;
        total += units[d];
    }
    bool any = false;
    for (int vp = 0; vp < a.nvol; vp++) {
        if (!a.neg[vp]) continue;  // (the same decision in every workgroup: the flags were written by earlier kernels)
        any = true;
        const int16_t* C = a.C + (size_t)vp * a.vol_stride;
        for (int u = blockIdx.x; u < total; u += gridDim.x) {
            int dir = 0, rest = u;
            while (rest >= units[dir]) rest -= units[dir++];
            scan_exact_unit<LANES, NR, STORE_SAT>(C, a.Lx, g, sd, dir, rest, a.min_as_int);
        }
        grid_sync();
        for (int y = blockIdx.x; y < g.H; y += gridDim.x) {
            wta_row<LANES, NR, true>(reinterpret_cast<const uint16_t*>(a.Lx), a.dst + (size_t)vp * a.dst_stride_e, a.dst_pitch_e,
                                     (size_t)0, g, (size_t)0, a.nd, sd.dir_stride, a.tie_lanes, a.combine, y, 0);
            __syncthreads();  // the row's LDS scratch is reused by the next row
        }
        grid_sync();
    }
    // a barrier that gave up: never hand back what may have been computed from half-written volumes
    if (any && (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u)) {
        for (int vp = 0; vp < a.nvol; vp++) {
            if (!a.neg[vp]) continue;
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.dst_n; i += (size_t)gridDim.x * 256)
                a.dst[(size_t)vp * a.dst_stride_e + i] = (int16_t)a.invalid;
        }
    }
}

}  // namespace camd

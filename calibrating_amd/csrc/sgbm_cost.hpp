// sgbm_cost.hpp -- the matching-cost volume in ONE pass: calcPixelCostBT + blockSize x blockSize box sum + P2
// -> C[y][x][d], written once (the split pair k_hsum / k_vsum of sgbm.hip moves the volume three times).
// Included by sgbm.hip (shares Geom).
//
//   C(y, x, d) = P2 + sum_{dy, dx in [-SW2, SW2]} pix(clamp(y + dy, 0, H-1), clamp(x + dx, 0, W1-1), d)
//
// Work decomposition ("lanes = columns, walk = rows"):
//   * a workgroup owns a strip of 64 consecutive cost columns (64 - (K-1) of them are outputs, K-1 the
//     horizontal halo), a block of up to 128 disparities and a chunk of rows that it walks top to bottom;
//   * lane l of every wave is column xs0 + l; wave w owns DL consecutive disparities, so a lane computes DL
//     pixel costs per row: its left-image operands are loaded once per row, the right-image operands of
//     consecutive disparities are consecutive LDS entries (lane stride = one entry: conflict-free b128 reads);
//   * the horizontal box sum is a trailing window over LANES: wave-wide DPP shifts (wave_shr:1) folded into
//     v_add_u32 on two packed u16 costs at a time (sums stay below 2^16, no carry between the halves);
//   * the vertical box sum is a running sum down the rows, the K rows of the window in a register ring
//     (the row loop is unrolled by K, every ring index is static);
//   * the BT operands of a row -- per image column (p, min(p, (p+l)/2, (p+r)/2), max(...)) with p = clipped
//     x-Sobel | raw << 16 -- are staged in LDS one row ahead of their use by a few "staging" waves: lane =
//     image column, the column's three rows fetched as unaligned dwords a step earlier, x-neighbours by DPP
//     wave shifts; double-buffered, so the row loop has ONE barrier per row.
// HBM traffic: the two images in, V out.
//
// SAT = true restates the int16 SATURATION of OpenCV's CV_SIMD build (v_int16 operator+ / operator- saturate)
// in the vertical recurrence, in OpenCV's operation order: column 0  C = (Cprev + hsumAdd) - hsumSub, columns
// >= 1  C = (Cprev - hsumSub) + hsumAdd, first row  C = P2 + (SH2+1)*h(0) + h(1) + ...; the recurrence then
// has to start at row 0 (one row chunk).  SAT = false wraps modulo 2^16 like the scalar build's (CostType)
// casts.  The two agree whenever K*K*cn*(2*ftzero + 63) + P2 <= 32767 -- and, image by image,
// whenever no value of the volume comes within one horizontal sum (K*cn*(2*ftzero + 63)) of 32767: then no
// intermediate of the saturating recurrence can clip either.  The host uses that (sgbm.hip): the chunked wrapping
// kernel runs first and reports per volume whether the bound held (ovf / ovf_thresh); the sequential saturating
// kernel, a single row chunk and therefore few workgroups, re-does only the volumes where it did not.
#pragma once

namespace camd {

static constexpr int COST_DL = 8;  // disparities per lane (the DL = 16 instantiations: two such octets, see k_cost)
// tunables (measured on MI355X)
#ifndef CAMD_COST_MAX_WAVES_RGB
#define CAMD_COST_MAX_WAVES_RGB 8    // waves per workgroup (x 8 disparities each)
#endif
#ifndef CAMD_COST_MAX_WAVES_GRAY
#define CAMD_COST_MAX_WAVES_GRAY 16
#endif
#ifndef CAMD_COST_MIN_WAVES
#define CAMD_COST_MIN_WAVES 6        // occupancy target (waves per SIMD) the register allocator works to
#endif
#ifndef CAMD_COST_MIN_WAVES_DL16
#define CAMD_COST_MIN_WAVES_DL16 4   // ... of the 16-disparities-per-lane form (K rows x 8 ring registers)
#endif
// The per-cell tail of the BT cost (sum over the colour planes with the raw planes >> 2, then two cells per register):
//   0  round 2-5: v_pk_lshrrev_b16 + v_dot2_u32_u16 per channel and cell, v_lshl_or + v_and per pair of cells --
//      all of them in the class that issues at ~0.9 per cycle and CU (tools/microtests/valu_rate.hip)
//   1  floor(x / 4) summed over the channels = (sum of (x & ~3)) >> 2: the mask rides in the v_bitop3_b32 that ORs the
//      two saturating differences anyway, the channels are summed with plain v_add_u32 (no half can carry:
//      3 * 255 < 2^16), and the gradient / raw halves of two cells are regrouped by two v_perm_b32 so that ONE plain
//      shift + add finishes both cells.  Per RGB cell: 7 "slow-class" instructions -> 1, + 3.5 plain ones.
#ifndef CAMD_COST_TAIL
#define CAMD_COST_TAIL 1
#endif
// the staging arithmetic: 0 = round 2-5 form, 1 = round 6 form (see stage_entries)
#ifndef CAMD_COST_STAGE
#define CAMD_COST_STAGE 1
#endif
#ifndef CAMD_COST_LDS_FLOOR_RGB
#define CAMD_COST_LDS_FLOOR_RGB (41 * 1024)
#endif
// How C leaves the kernel.  A pixel's disparity vector (256 B at D = 128) is produced 16 bytes at a time by the waves of
// one or two workgroups; stored straight from the registers, every wave's store instruction touches 64 different lines
// with 16 bytes each and the L2 has to combine 8 or 16 such pieces per line -- for gray, whose arithmetic is a third of
// RGB's, that is what bounds the kernel (V written at 3.0 TB/s, 0.7 partial-line writes per clock and L2 channel).
// With the bit set (bit 0 gray, bit 1 RGB) a row's 64 x NW pieces go through a double-buffered LDS tile instead
// (written behind the row's arithmetic, read back after the row's barrier) and every wave stores whole runs of
// NW x 16 contiguous bytes per pixel: the same number of store instructions, 1/NW of the L2 write transactions.
#ifndef CAMD_COST_TSTORE
#define CAMD_COST_TSTORE 1
#endif
// 1 (measurement build): the C stores carry `nt`.  A pixel's 256-byte disparity vector is written 16 bytes at a time by
// 16 waves of two workgroups; streamed past the L2 those pieces reach HBM as partial lines: 72 instead of 16 ms
// (profiles/r06_band_nt.txt).  The L2's write combining is what makes the "lanes = columns" store pattern affordable.
#ifndef CAMD_COST_NT
#define CAMD_COST_NT 0
#endif
// 1: RGB at blockSize <= 5 runs 16 disparities per lane (sgbm.hip: the launch); 2: gray too; 0: 8 everywhere
#ifndef CAMD_COST_DL16
#define CAMD_COST_DL16 0
#endif

// n applications of the one-lane wave shift (lane i <- lane i-1, lane 0 <- 0)
template <int N>
__device__ __forceinline__ uint32_t wave_shr(uint32_t v)
{
#pragma unroll
    for (int i = 0; i < N; i++) v = dpp_perm<DPP_WAVE_SHR1>(v);
    return v;
}

// trailing window sum over lanes: out(l) = p(l) + p(l-1) + ... + p(l-K+1), by doubling
template <int K>
__device__ __forceinline__ uint32_t lane_window_sum(uint32_t p)
{
    if (K == 1) return p;
    const uint32_t w2 = p + wave_shr<1>(p);
    if (K == 3) return p + wave_shr<1>(w2);
    if (K == 5) {
        const uint32_t w4 = w2 + wave_shr<2>(w2);
        return p + wave_shr<1>(w4);
    }
    if (K == 7) {
        const uint32_t w3 = p + wave_shr<1>(w2);
        const uint32_t w6 = w3 + wave_shr<3>(w3);
        return p + wave_shr<1>(w6);
    }
    if (K == 9) {
        const uint32_t w4 = w2 + wave_shr<2>(w2);
        const uint32_t w8 = w4 + wave_shr<4>(w4);
        return p + wave_shr<1>(w8);
    }
    // K == 11
    const uint32_t w4 = w2 + wave_shr<2>(w2);
    const uint32_t w5 = p + wave_shr<1>(w4);
    const uint32_t w10 = w5 + wave_shr<5>(w5);
    return p + wave_shr<1>(w10);
}

// The same window as a difference of wave-wide inclusive prefix sums: six DPP adds (row_shr 1, 2, 4, 8, row_bcast 15 / 31)
// + one ds_bpermute for P(l - K) instead of K - 1 single-lane shifts.  For K >= 9 (the reference's block 11: ten
// shifts + five adds per packed register).  The halves cannot carry: a prefix over 64 lanes of pixel costs is at most
// 64 * CN * (2 * CAMD_MAX_FTZERO + 63) < 65536.  `back` = byte address of lane l - K for ds_bpermute, `live` = l >= K.
// Measured in round 5 on the reference's default matcher (block 11 x RGB, 64 pairs of 1000 x 562): 9.7 ms either way
// (bit-exact; tools/gpu_default_batch.py) -- at block 11 the kernel is not bound by the count of its window operations
// (four waves per SIMD at 128 VGPRs for the 11-row ring, 27 ds_read_b128 per row).  Off by default.
#ifndef CAMD_COST_SCAN_WINDOW
#define CAMD_COST_SCAN_WINDOW 0
#endif
static_assert(64 * 3 * (2 * CAMD_MAX_FTZERO + 63) < 65536, "a 64-lane prefix of pixel costs must fit a 16-bit half");
__device__ __forceinline__ uint32_t lane_window_sum_scan(uint32_t p, int back, bool live)
{
    uint32_t s = p;
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xf, 0xf, false);  // row_shr:1
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xf, 0xf, false);  // row_shr:2
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xf, false);  // row_shr:4
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xf, false);  // row_shr:8
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    const uint32_t prev = (uint32_t)__builtin_amdgcn_ds_bpermute(back, (int)s);
    return s - (live ? prev : 0u);
}

// Row ranges the cost volume is built for.  Every mode but MODE_SGBM_3WAY has ONE range, the image; 3WAY has one per
// stripe (cv2 computes its stripes independently: the vertical box window is clamped at the stripe's first row, and
// the volume of a stripe is stored as a "virtual pair" of its own, so that the aggregation kernels see independent
// images).  start = image row of local row 0 (also the lower clamp of the window), rows = rows in the range.
struct CostRanges {
    int n;
    int start[4], rows[4];
};

// C leaves the kernel through LDS (CAMD_COST_TSTORE bit 0: gray, bit 1: RGB) when the waves of the workgroup divide the
// strip's 64 columns: see the store in cost_body
constexpr bool cost_tstore(int cn) { return (CAMD_COST_TSTORE & (cn == 1 ? 1 : 2)) != 0; }
static inline bool cost_tstore_shape(int cn, int nwaves, int dl) { return cost_tstore(cn) && dl == COST_DL && 64 % nwaves == 0; }
static inline size_t cost_lds_bytes(int cn, int nwaves, int dl = COST_DL)
{
    const int es = cn == 1 ? 4 : 12, dw = nwaves * dl;
    const int nr = 64 + dw - 1, nl = 64;
    const size_t need = (size_t)2 * (nr + nl) * es * 4 + (cost_tstore_shape(cn, nwaves, dl) ? (size_t)2 * 64 * (nwaves + 1) * 16 : 0);
    // RGB at 8 waves: the kernel needs 64 VGPRs, so FOUR workgroups would fit a CU and fill every wave slot -- which leaves
    // the other batch in flight (bench.py's second stream) no room beside it.  Asking for a third of the LDS keeps it at three.
    return (cn == 3 && nwaves == 8 && need < CAMD_COST_LDS_FLOOR_RGB) ? (size_t)CAMD_COST_LDS_FLOOR_RGB : need;
}

// Only one dword of an entry's third quad is used; left alone the compiler narrows that read to ds_read_b32, whose 32
// lanes at a 48-byte stride collide four ways on the LDS banks, while the full ds_read_b128 is conflict-free
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define KEEP_B128(q) asm volatile("" ::"v"(q))

// The register ring of the vertical box sum holds K rows of 4 registers: from blockSize 9 on the kernel does not fit the
// 80 registers of six waves per SIMD (blockSize 11 RGB, the reference's default: 40 spilled, 10.5 ms per 64 pairs of
// 1000 x 562, D = 218); allowed 96 / 128 it runs without scratch traffic on fewer waves: 9.7 / 9.6 ms.
constexpr int cost_min_waves(int K, int DL = COST_DL)
{
    return DL > 8 ? CAMD_COST_MIN_WAVES_DL16 : (K >= 11 ? 4 : (K >= 9 ? 5 : CAMD_COST_MIN_WAVES));
}

// the work of one (strip bx, row chunk x disparity block by, volume bz) -- k_cost's workgroup, or one item of a
// persistent workgroup (k_cost_persist)
template <int CN, int K, bool SAT, int DL>
__device__ __forceinline__ void cost_body(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                          size_t pitch, size_t image_stride, uint16_t* __restrict__ Cout,
                                          const Geom& g, int rb, int nchunks, size_t vol_stride, const CostRanges& cr,
                                          uint32_t* __restrict__ ovf, int ovf_thresh, uint32_t* __restrict__ neg,
                                          const int bx, const int by, const int bz)
{
    constexpr int ES = CN == 1 ? 4 : 12;  // dwords per staged entry: (p, lo, hi) per channel, padded to 16 bytes
    constexpr int EV = ES / 4;
    constexpr int NP = DL / 2;            // packed cost registers per lane
    constexpr int SW2 = K / 2;
    constexpr int XS = 64 - (K - 1);      // output columns per strip
    // every pixel cost is at most 2*ftzero + 63 per channel; ftzero and P2 are bounded by the limits normalise() enforces
    constexpr bool NOCARRY = K * K * CN * (2 * CAMD_MAX_FTZERO + 63) + CAMD_MAX_P2 <= 65535;
    extern __shared__ uint4 cs_lds[];  // everything in LDS is addressed in 16-byte quads: b128 reads and writes

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6, DW = NW * DL;
    const int tid = threadIdx.x;
    const int chunk = by % nchunks, dblk = by / nchunks;
    // (the division runs on the vector unit: say that its results are uniform, or everything derived from them -- image
    // pointers, buffer descriptors, row addresses -- is treated as per-lane)
    const int vpair = bz, pair = __builtin_amdgcn_readfirstlane(vpair / cr.n), ridx = __builtin_amdgcn_readfirstlane(vpair % cr.n);  // volume index, image index, range
    // (indexing the by-value CostRanges with ridx goes through vector registers: tell the compiler the result is uniform, or
    // every row address downstream is computed per lane in 64-bit vector arithmetic)
    const int ybase = __builtin_amdgcn_readfirstlane(cr.start[ridx]), nrows = __builtin_amdgcn_readfirstlane(cr.rows[ridx]);
    if (chunk * rb >= nrows) return;  // (ranges shorter than the longest one: whole workgroup, before any barrier)
    // Two-stage build of a saturating volume (sgbm.hip): the wrapping kernel (SAT = false, rows in parallel chunks)
    // raises ovf[volume] when a value it writes exceeds ovf_thresh; this sequential kernel then only runs for the
    // volumes whose flag is up
    if (SAT && ovf && !ovf[vpair]) return;
    uint32_t ovf_max = 0;
    // SAT also reports (neg[volume]) whether a value it wrote is below P2: after a clipped sum the recurrence keeps what
    // it lost, and a volume with C < P2 is outside the regime of the packed-u16 aggregation kernels (sgbm_exact.hpp)
    uint32_t neg_min = SENT_PK;
    const int W1 = g.W1, H = g.H;
    const int xo0 = bx * XS, xs0 = xo0 - SW2;                 // first output column, column of lane 0
    const int cmin = max(xs0, 0), cmax = min(xs0 + 63, W1 - 1);       // clamped column range of the strip
    const int cx = min(max(xs0 + lane, 0), W1 - 1);                   // this lane's (clamped) cost column
    const int db = dblk * DW;                                         // first disparity index of the block
    const int NL = cmax - cmin + 1, NR = NL + DW - 1;                 // staged left / right image columns
    const int lcol0 = cmin + g.minX1;                                 // image column of left entry 0
    const int rcol0 = cmin + g.minX1 - g.minD - (db + DW - 1);        // image column of right entry 0
    const int NRmax = 64 + DW - 1, NLmax = 64;
    const int esz = (NRmax + NLmax) * EV;  // quads per buffer
    uint4* const Ebuf = cs_lds;            // [2][esz]: right entries, then left entries

    const uint8_t* imgL = left + (size_t)pair * image_stride;
    const uint8_t* imgR = right + (size_t)pair * image_stride;
    const int ftz = g.ftzero;
    const uint32_t ftz2 = (uint32_t)ftz | ((uint32_t)ftz << 16);

    // rows (local to the range): step r of the walk handles image row clamp(ybase + y0 - SW2 + r, ybase, H-1); output
    // row y0 + r - (K-1)
    const int y0 = chunk * rb, y1 = min(y0 + rb, nrows);
    const int nsteps = (y1 - y0) + K - 1;
    auto row_of = [&](int r) { return min(max(ybase + y0 - SW2 + r, ybase), H - 1); };

    // ---- staging: one lane = one image column, neighbours by wave-wide DPP shifts ---------------------------------
    // The staged columns (NR of the right image, NL of the left) are cut into pieces of <= 60 columns; a piece sits
    // in consecutive lanes of ONE wave with two extra columns on each side (p needs the vertical sums of x-1, x+1;
    // the entry needs p of x-1, x+1).  Every lane finds its piece once; waves without a piece skip the staging.
    int st_k = 0, st_lo = 0, st_hi = 0, st_img = -1, st_total = 0;  // st_total: threads that hold a staging lane
    {
        int v = 0;
        for (int sg = 0; sg < 2; sg++) {
            const int n = sg ? NL : NR;
            for (int e = 0; e < n;) {
                int room = 64 - (v & 63);
                if (room < 5) { v += room; room = 64; }
                const int len = min(n - e, room - 4);
                if (tid >= v && tid < v + len + 4) { st_img = sg; st_k = e + (tid - v) - 2; st_lo = e; st_hi = e + len; }
                v += len + 4;
                e += len;
            }
        }
        st_total = v;
    }
    // A wave whose DL disparities are all padding (d >= D: the tail of numDisparities rounded up to the volume's layout,
    // 38 of 256 at the reference's D = 218) and that holds no staging lane has nothing to do and leaves before the first
    // barrier (the hardware counts only live waves at s_barrier).  Nothing reads the padded part of C for its value --
    // every consumer masks d >= D -- and camd_sgbm_create filled it with P2 once, which is what this wave would write.
    if (db + w * DL >= g.D && w * 64 >= st_total) return;
    const bool stager = st_img >= 0;
    const bool st_store = stager && st_k >= st_lo && st_k < st_hi;
    const int st_col = (st_img == 1 ? lcol0 : rcol0) + st_k;
    const bool st_inside = st_col > 0 && st_col < g.W - 1;  // OpenCV: columns 0 and W-1 of every plane hold tab[0]
    const int st_ccol = min(max(st_col, 0), g.W - 1);
    // RGB pixels are fetched as ONE unaligned dword (R | G << 8 | B << 16 | next byte); the last column of a row is
    // fetched one byte early and shifted so that no load reaches past the row
    const bool st_last = CN == 3 && st_ccol == g.W - 1;
    // byte offset of this lane's column inside an image row (the row itself is a scalar: see fetch_rows)
    const uint32_t st_off = (uint32_t)(st_ccol * CN - (st_last ? 1 : 0));
    const uint32_t st_shift = st_last ? 8u : 0u;
    uint4* const st_dst = Ebuf + (st_img == 1 ? NRmax * EV : 0) + st_k * EV;  // + buffer * esz

    uint32_t rowA = 0, rowB = 0, rowC = 0;  // the three image rows of the column being staged (in flight)
    auto fetch_rows = [&](int y) {
        if (stager) {
            // buffer loads: descriptor = the image ROW (a uniform 64-bit address the scalar unit computes; raw, byte-addressed),
            // voffset = this lane's column -- no vector arithmetic per row (global loads from a per-lane 64-bit pointer
            // cost three v_mad_u64_u32 + three 64-bit adds per staged row)
            const size_t om = (size_t)(y > 0 ? y - 1 : y) * pitch, o0 = (size_t)y * pitch, op = (size_t)(y < H - 1 ? y + 1 : y) * pitch;
            auto row_of_img = [&](const uint8_t* row) -> uint32_t {  // (row: uniform; the descriptor is scalar arithmetic)
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(row), 0, 0x7fffffff, 0x00020000);
                if (CN == 1) return __builtin_amdgcn_raw_buffer_load_b8(rs, st_off, 0, 0);
                return __builtin_amdgcn_raw_buffer_load_b32(rs, st_off, 0, 0);
            };
            auto rows_of = [&](const uint8_t* img) {
                rowA = row_of_img(img + om);
                rowB = row_of_img(img + o0);
                rowC = row_of_img(img + op);
            };
            // (one wave of the four holds lanes of both images and runs both branches.  The empty asm statements keep the
            // compiler from merging the branches into ONE load through a per-lane select of the descriptor, which it then
            // has to serialise with a readfirstlane loop)
            if (st_img == 1) {
                rows_of(imgL);
                asm volatile("; left image rows" ::: "memory");
            } else {
                rows_of(imgR);
                asm volatile("; right image rows" ::: "memory");
            }
        }
    };
    // The entry arithmetic, round 6 form (CAMD_COST_STAGE 1): everything that can be a plain 32-bit operation is one
    // (tools/microtests/valu_rate.hip: they issue at ~1.55 per cycle and CU, the packed / DPP / three-operand forms at
    // ~0.9).  Vertical Sobel part on the even and the odd bytes of the pixel dwords, two channels per register:
    //     sE = (s0 | s2 << 16),  sO = (s1 | junk << 16),   s = I(y-1) + 2 I(y) + I(y+1) <= 1020
    // gradient s(x+1) - s(x-1) with a bias of 1024 per half so that the plain subtraction cannot borrow, clipped by a
    // packed max / min against (1024 -+ ftz), re-based by a plain subtraction; one v_perm_b32 per channel pairs it with
    // the raw byte.  The half-pixel interval uses  min(u, (u+l)/2, (u+r)/2) = (u + min(u, l, r)) / 2  (t -> (u+t)/2 is
    // monotone; likewise max), the halving sum of two 8-bit values in 16-bit fields being one v_lerp_u8.
    // Per staged RGB column 35 of the slower class + 20 plain instead of 55 + 11 (round 2-5 form: CAMD_COST_STAGE 0).
    const uint32_t st_keep = st_inside ? 0xffffffffu : 0u, st_fill = st_inside ? 0u : ftz2;
    const uint32_t clip_lo = dup16(1024u - (uint32_t)ftz), clip_hi = dup16(1024u + (uint32_t)ftz);
    auto stage_entries = [&](int buf) {
        if (stager) {  // wave-uniform up to the last staging wave
            uint32_t u[CN], lo[CN], hi[CN];
            const uint32_t a = rowA >> st_shift, b = rowB >> st_shift, c = rowC >> st_shift;
            if (CAMD_COST_STAGE == 0) {
#pragma unroll
                for (int ch = 0; ch < CN; ch++) {
                    // vertical part of the x-Sobel: s = I(y-1) + 2 I(y) + I(y+1) -- the three rows of this channel
                    // gathered into one dword, then one dot product with (1, 2, 1)
                    const uint32_t t = __builtin_amdgcn_perm(b, a, 0x0c0c0400u + 0x00000101u * ch);      // (a.ch, b.ch, 0, 0)
                    const uint32_t t3 = __builtin_amdgcn_perm(c, t, 0x0c040100u + 0x00010000u * ch);     // (a.ch, b.ch, c.ch, 0)
                    const uint32_t sv = __builtin_amdgcn_udot4(t3, 0x00010201u, 0u, false);
                    // gradient = s(x+1) - s(x-1), clipped to [-ftz, ftz], + ftz;  p = gradient | raw << 16
                    // (the subtrahend passes through an empty asm so that the DPP move is NOT folded into the subtraction:
                    // the folded form, v_subrev_u32_dpp, measured wrong on gfx950 -- it returned shr(src1) - src0)
                    uint32_t sl = dpp_perm<DPP_WAVE_SHR1>(sv);
                    asm volatile("" : "+v"(sl));
                    const int gq = (int)dpp_perm<DPP_WAVE_SHL1>(sv) - (int)sl;
                    const uint32_t gc = (uint32_t)(min(max(gq, -ftz), ftz) + ftz);
                    const uint32_t raw = (b >> (8 * ch)) & 0xffu;
                    u[ch] = st_inside ? (gc | (raw << 16)) : ftz2;
                }
#pragma unroll
                for (int ch = 0; ch < CN; ch++) {
                    // half-pixel interval: columns outside the image carry ftz2 like the border columns, so the
                    // "no neighbour at the image edge" rule (use p itself) needs no special case
                    const uint32_t l = dpp_perm<DPP_WAVE_SHR1>(u[ch]), r = dpp_perm<DPP_WAVE_SHL1>(u[ch]);
                    const uint32_t ul = pk_lshr_u16(pk_add_u16(u[ch], l), 0x00010001u);
                    const uint32_t ur = pk_lshr_u16(pk_add_u16(u[ch], r), 0x00010001u);
                    lo[ch] = pk_min_u16(pk_min_u16(ul, ur), u[ch]);
                    hi[ch] = pk_max_u16(pk_max_u16(ul, ur), u[ch]);
                }
            } else {
                uint32_t vc[2];  // clipped gradients + ftz: (ch0 | ch2 << 16), (ch1 | junk << 16)
#pragma unroll
                for (int eo = 0; eo < (CN == 1 ? 1 : 2); eo++) {
                    const uint32_t M = CN == 1 ? 0xffu : 0x00ff00ffu;
                    const uint32_t ea = (eo ? a >> 8 : a) & M, eb = (eo ? b >> 8 : b) & M, ec = (eo ? c >> 8 : c) & M;
                    const uint32_t sv = ea + ec + eb + eb;
                    // (the subtrahend passes through an empty asm so that its DPP move is NOT folded into the
                    // subtraction: the folded form, v_subrev_u32_dpp, measured wrong on gfx950)
                    uint32_t sl = dpp_perm<DPP_WAVE_SHR1>(sv);
                    asm volatile("" : "+v"(sl));
                    const uint32_t gq = dpp_perm<DPP_WAVE_SHL1>(sv) - sl + 0x04000400u;  // 1024 + gradient per half
                    vc[eo] = pk_min_u16(pk_max_u16(gq, clip_lo), clip_hi) - clip_lo;
                }
#pragma unroll
                for (int ch = 0; ch < CN; ch++) {
                    // (gradient of the channel | its raw byte << 16); columns 0 and W-1 (and beyond) hold tab[0]
                    const uint32_t sel = ch == 0 ? 0x0c040100u : (ch == 1 ? 0x0c050100u : 0x0c060302u);
                    u[ch] = (__builtin_amdgcn_perm(b, vc[ch & 1], sel) & st_keep) | st_fill;
                }
#pragma unroll
                for (int ch = 0; ch < CN; ch++) {
                    const uint32_t l = dpp_perm<DPP_WAVE_SHR1>(u[ch]), r = dpp_perm<DPP_WAVE_SHL1>(u[ch]);
                    lo[ch] = __builtin_amdgcn_lerp(u[ch], pk_min_u16(pk_min_u16(l, r), u[ch]), 0u);
                    hi[ch] = __builtin_amdgcn_lerp(u[ch], pk_max_u16(pk_max_u16(l, r), u[ch]), 0u);
                }
            }
            if (st_store) {
                u32x4_t* d4 = reinterpret_cast<u32x4_t*>(st_dst + buf * esz);
                if (CN == 1) {
                    d4[0] = u32x4_t{u[0], lo[0], hi[0], 0u};
                } else {
                    d4[0] = u32x4_t{u[0], lo[0], hi[0], u[1 % CN]};
                    d4[1] = u32x4_t{lo[1 % CN], hi[1 % CN], u[2 % CN], lo[2 % CN]};
                    d4[2] = u32x4_t{hi[2 % CN], 0u, 0u, 0u};
                }
            }
        }
    };

    fetch_rows(row_of(0));
    stage_entries(0);
    fetch_rows(row_of(1));  // consumed at the end of step 0
    __syncthreads();

    // ---- per-lane constants -----------------------------------------------------------------------------------
    const int d0 = db + w * DL;                               // first disparity index of this wave
    const int d0u = db + __builtin_amdgcn_readfirstlane(w) * DL;  // (the same, known to be wave-uniform: SGPR operands)
    uint32_t keep[NP];                                        // padded disparities d >= D carry pix = 0 (C = P2)
#pragma unroll
    for (int k = 0; k < NP; k++)
        keep[k] = (d0u + 2 * k < g.D ? 0xffffu : 0u) | (d0u + 2 * k + 1 < g.D ? 0xffff0000u : 0u);
    const uint32_t rawmask = 0xfffcffffu;                     // (gradient | raw << 16): raw rounded down to 4 n
    int eoff_l = NRmax * EV + (cx - cmin) * EV;                                    // uint4 index of the left entry
    int eoff_r = ((cx - cmin) + DW - 1 - w * DL - (DL - 1)) * EV;                  // right entry of cell DL-1
    // opaque to the optimiser: the per-cell entries are then reached with non-negative immediate offsets from this
    // base (re-associated, the lowest address would be a negative offset and cost a VALU add per LDS read)
    asm volatile("" : "+v"(eoff_l), "+v"(eoff_r));
    const int xo = xo0 + lane - (K - 1);                                           // output column of this lane
    const bool writer = lane >= K - 1 && xo < W1 && d0 < g.Dp;
    const bool first_col = xo == 0;
    uint16_t* const vol = Cout + (size_t)vpair * vol_stride;                 // (uniform)
    const uint32_t out_off = (uint32_t)((writer ? xo : 0) * g.Dp + d0);     // element offset of this lane inside a row of C

    const int win_back = ((lane - K) & 63) << 2;
    const bool win_live = lane >= K;
    const uint32_t p2 = dup16((uint32_t)g.P2);
    // the output tile (see CAMD_COST_TSTORE): [2][64 columns][NW + 1] quads behind the entry buffers; this lane writes
    // quad (lane, w) and later stores the quad (column tpx, piece tpc) -- NW consecutive lanes = one column's NW pieces
    // (uniform; the launch sized the LDS the same way.  Not for a disparity block with padding: its all-padding waves have
    // left, and with them the columns they would flush)
    const bool tstore = cost_tstore(CN) && DL == COST_DL && 64 % NW == 0 && db + DW <= g.D;
    const int TP = NW + 1;
    uint4* const Tbuf = Ebuf + 2 * esz;
    const int tpx = w * (64 / NW) + lane / NW, tpc = lane % NW;
    const int txo = xo0 + tpx - (K - 1);
    const bool twriter = tpx >= K - 1 && txo < W1 && db + tpc * DL < g.Dp;
    const uint32_t tout_off = (uint32_t)((twriter ? txo : 0) * g.Dp + db + tpc * DL);
    // a 16-byte store into row `row` of C (uniform address) at element offset `off` of this lane: buffer store, the row in the
    // descriptor (scalar arithmetic), the lane's part a 32-bit vector offset
    auto store_c = [&](uint16_t* row, uint32_t off, u32x4_t v) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off * 2u, 0, 0);
    };
    auto flush_row = [&](int r) {  // row step r's tile -> C (after the barrier that ended step r)
        if (twriter) {
            const uint4 v = Tbuf[((r & 1) * 64 + tpx) * TP + tpc];
            store_c(vol + (size_t)(y0 + r - (K - 1)) * W1 * g.Dp, tout_off, u32x4_t{v.x, v.y, v.z, v.w});
        }
    };
    uint32_t acc[NP], ring[K][NP];
#pragma unroll
    for (int k = 0; k < NP; k++) acc[k] = p2;
#pragma unroll
    for (int u = 0; u < K; u++)
#pragma unroll
        for (int k = 0; k < NP; k++) ring[u][k] = 0;

    for (int r0 = 0; r0 < nsteps; r0 += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int r = r0 + u;
            if (r < nsteps) {  // uniform
                if (tstore && r >= K) flush_row(r - 1);
                const uint4* E4 = Ebuf + (r & 1) * esz;
                uint32_t U[CN], U0[CN], U1[CN];
                {
                    const u32x4_t* q = reinterpret_cast<const u32x4_t*>(E4 + eoff_l);
                    if (CN == 1) {
                        const u32x4_t a = q[0];
                        U[0] = a.x; U0[0] = a.y; U1[0] = a.z;
                    } else {
                        const u32x4_t a = q[0], b = q[1], c = q[2];
                        KEEP_B128(c);
                        U[0] = a.x; U0[0] = a.y; U1[0] = a.z;
                        U[1 % CN] = a.w; U0[1 % CN] = b.x; U1[1 % CN] = b.y;
                        U[2 % CN] = b.z; U0[2 % CN] = b.w; U1[2 % CN] = c.x;
                    }
                }
                uint32_t cost[DL];
#pragma unroll
                for (int j = 0; j < DL; j++) {
                    const u32x4_t* q = reinterpret_cast<const u32x4_t*>(E4 + eoff_r + (DL - 1 - j) * EV);
                    uint32_t V[CN], V0[CN], V1[CN];
                    if (CN == 1) {
                        const u32x4_t a = q[0];
                        V[0] = a.x; V0[0] = a.y; V1[0] = a.z;
                    } else {
                        const u32x4_t a = q[0], b = q[1], c = q[2];
                        KEEP_B128(c);
                        V[0] = a.x; V0[0] = a.y; V1[0] = a.z;
                        V[1 % CN] = a.w; V0[1 % CN] = b.x; V1[1 % CN] = b.y;
                        V[2 % CN] = b.z; V0[2 % CN] = b.w; V1[2 % CN] = c.x;
                    }
                    uint32_t a32 = 0;
#pragma unroll
                    for (int c = 0; c < CN; c++) {
                        // c0 = max(0, u - v1, v0 - u), c1 = max(0, v - u1, u0 - v): at most one term of each pair
                        // is non-zero, so OR of the saturating differences is their max
                        if (CAMD_COST_TAIL == 0) {
                            const uint32_t a = pk_subsat_u16(U[c], V1[c]) | pk_subsat_u16(V0[c], U[c]);
                            const uint32_t b = pk_subsat_u16(V[c], U1[c]) | pk_subsat_u16(U0[c], V[c]);
                            uint32_t m = pk_min_u16(a, b);
                            m = pk_lshr_u16(m, 0x00020000u);  // raw plane: cost >> 2
                            a32 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, m),
                                                         __builtin_bit_cast(u16x2_t, 0x00010001u), a32, false);
                        } else {
                            // the raw half rounded down to a multiple of 4 (min of the rounded = the rounded min); the
                            // mask is the third input of the v_bitop3_b32 that ORs the differences
                            const uint32_t a = (pk_subsat_u16(U[c], V1[c]) | pk_subsat_u16(V0[c], U[c])) & rawmask;
                            const uint32_t b = (pk_subsat_u16(V[c], U1[c]) | pk_subsat_u16(U0[c], V[c])) & rawmask;
                            a32 += pk_min_u16(a, b);  // gradient sum | raw sum << 16, plain add
                        }
                    }
                    cost[j] = a32;
                }
                // pack two disparities per register
                uint32_t pp[NP];
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    if (CAMD_COST_TAIL == 0) {
                        pp[k] = (cost[2 * k] | (cost[2 * k + 1] << 16)) & keep[k];
                    } else {
                        const uint32_t gq = __builtin_amdgcn_perm(cost[2 * k + 1], cost[2 * k], 0x05040100u);  // gradient sums
                        const uint32_t rq = __builtin_amdgcn_perm(cost[2 * k + 1], cost[2 * k], 0x07060302u);  // raw sums (x 4)
                        pp[k] = (gq + (rq >> 2)) & keep[k];  // bits 16, 17 of rq are zero: the plain shift is clean
                    }
                }
                // horizontal window over lanes, vertical running sum
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const uint32_t T = (CAMD_COST_SCAN_WINDOW && K >= 9) ? lane_window_sum_scan(pp[k], win_back, win_live)
                                                                         : lane_window_sum<K>(pp[k]);
                    const uint32_t old = ring[u][k];
                    ring[u][k] = T;
                    if (SAT) {
                        // (Cprev + add) - sub in column 0 while the entering row exists (y + SH2 < H), else and
                        // everywhere else (Cprev - sub) + add
                        const uint32_t a0 = pk_subsat_i16(pk_addsat_i16(acc[k], T), old);
                        const uint32_t a1 = pk_addsat_i16(pk_subsat_i16(acc[k], old), T);
                        acc[k] = (first_col && ybase + y0 + r - (K - 1) + SW2 < H) ? a0 : a1;
                    } else {
                        // plain 32-bit add / subtract (twice the issue rate of the packed forms) where no half can
                        // carry: the running sum is a true sum of at most K*K*CN pixel costs + P2
                        acc[k] = NOCARRY ? acc[k] + T - old : pk_sub_u16(pk_add_u16(acc[k], T), old);
                    }
                }
                if (tstore && r >= K - 1) Tbuf[((r & 1) * 64 + lane) * TP + w] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                if (r >= K - 1 && writer) {
                    const int y = y0 + r - (K - 1);
                    uint16_t* const orow = vol + (size_t)y * W1 * g.Dp;
#pragma unroll
                    for (int q = 0; q < NP / 4; q++) {
                        if (tstore) {
                        } else if (CAMD_COST_NT)
                            __builtin_nontemporal_store(u32x4_t{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]},
                                                        reinterpret_cast<u32x4_t*>(orow + out_off) + q);
                        else
                            store_c(orow, out_off + 8 * q, u32x4_t{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
                        if (SAT) neg_min = pk_min_i16(pk_min_i16(neg_min, pk_min_i16(acc[4 * q], acc[4 * q + 1])),
                                                      pk_min_i16(acc[4 * q + 2], acc[4 * q + 3]));
                        if (!SAT && ovf_thresh >= 0)  // (uniform)
                            ovf_max = pk_max_u16(pk_max_u16(ovf_max, pk_max_u16(acc[4 * q], acc[4 * q + 1])),
                                                 pk_max_u16(acc[4 * q + 2], acc[4 * q + 3]));
                    }
                }
                // entries of row r+1 from the rows fetched one step ago; then fetch for row r+2
#if defined(CAMD_COST_DBG_NOSTAGE) && !defined(CAMD_MEASUREMENT_BUILD)
#error "CAMD_COST_DBG_NOSTAGE produces wrong results: measurement builds only (define CAMD_MEASUREMENT_BUILD too)"
#endif
#ifndef CAMD_COST_DBG_NOSTAGE  // (measurement only: what the staging costs; the results are wrong without it)
                stage_entries((r + 1) & 1);
                fetch_rows(row_of(r + 2));
#endif
                __syncthreads();
            }
        }
    }
    if (tstore) flush_row(nsteps - 1);  // (the last step's barrier is behind us)
    if (!SAT && ovf_thresh >= 0 && (int)max(ovf_max & 0xffffu, ovf_max >> 16) > ovf_thresh) atomicOr(ovf + vpair, 1u);
    if (SAT && neg && min((int)(int16_t)(neg_min & 0xffffu), (int)(int16_t)(neg_min >> 16)) < g.P2) atomicOr(neg + vpair, 1u);
}

template <int CN, int K, bool SAT, int DL = COST_DL>
__global__ __launch_bounds__(1024, cost_min_waves(K, DL)) void k_cost(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                               size_t pitch, size_t image_stride, uint16_t* __restrict__ Cout,
                                               Geom g, int rb, int nchunks, size_t vol_stride, CostRanges cr,
                                               uint32_t* __restrict__ ovf, int ovf_thresh, uint32_t* __restrict__ neg)
{
    cost_body<CN, K, SAT, DL>(left, right, pitch, image_stride, Cout, g, rb, nchunks, vol_stride, cr, ovf, ovf_thresh, neg,
                              (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// The same work handed out by a ticket to a FIXED number of resident workgroups (grid = workgroups per CU x 256): a
// launch that does not fill the chip, so that the workgroups of another kernel -- the HBM-bound last aggregation pass of
// the previous batch, launched the same way on another stream -- are resident beside it for its whole duration.  (Two
// ordinary launches on two streams only overlap in their tails: the dispatcher drains the older grid first --
// profiles/r06_corun.json.)  Items are numbered strip-fastest like k_cost's grid.  Only for launches in which no wave
// leaves early (numDisparities a multiple of the workgroup's disparity block) and no per-volume early exit applies.
template <int CN, int K, int DL = COST_DL>
__global__ __launch_bounds__(1024, cost_min_waves(K, DL)) void k_cost_persist(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                               size_t pitch, size_t image_stride, uint16_t* __restrict__ Cout,
                                               Geom g, int rb, int nchunks, size_t vol_stride, CostRanges cr,
                                               uint32_t* __restrict__ ticket, int nx, int ny, int nitems)
{
    __shared__ int s_item;
    for (;;) {
        if (threadIdx.x == 0) s_item = (int)atomicAdd(ticket, 1u);
        __syncthreads();
        const int item = s_item;
        __syncthreads();
        if (item >= nitems) break;
        const int bx = item % nx, r = item / nx;
        cost_body<CN, K, false, DL>(left, right, pitch, image_stride, Cout, g, rb, nchunks, vol_stride, cr, nullptr, -1, nullptr,
                                    bx, r % ny, r / ny);
    }
}

}  // namespace camd

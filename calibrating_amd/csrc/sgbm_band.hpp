// sgbm_band.hpp -- fused multi-direction aggregation for gfx950: the "band wavefront" pass.
// Included by sgbm.hip (shares Geom, SENT_PK, KEY_INIT).
//
// One pass aggregates the FOUR directions that share a sweep orientation (sx, sy) -- OpenCV's own grouping:
//     H = (sx, 0)      V = (0, sy)      Dg = (sx, sy)      A = (-sx, sy)
// so that the cost volume C is read once and S is written (first pass) or read (last pass) once for all of
// them: 2V of HBM traffic per pass instead of 3V per direction.
//     MODE_HH   = 2 passes  {->, v, \, /}  and  {<-, ^, \^, /^} + WTA
//     MODE_SGBM = {->, v, \, /}  and a row-parallel {<-} + WTA pass (FULL = false: rows are independent, so
//                 there is no exchange, no helper wave and no barrier)
//
// Work decomposition: the image is cut into bands of R = BAND_THREADS/LANES rows.  A workgroup owns a
// band; its group g (LANES lanes, one pixel's disparity vector) owns row g of the band and walks it in
// sweep order, skewed by TWO columns per row: at step t group g is at column index xi = t - 2g.  Then
//     H  input (xi-1, row)    = the group's own registers,
//     A  input (xi+1, row-1)  = what group g-1 produced in step t-1   (used at once),
//     V  input (xi,   row-1)  = produced in step t-2                   (read at t-1, held one step),
//     Dg input (xi-1, row-1)  = produced in step t-3                   (read at t-2, held two steps),
// i.e. the consumer reads the producer's three vectors of column xi+1 once per step from a double-buffered
// LDS slot per group (ONE LDS-only barrier per step) and keeps V / Dg in register sets that rotate with
// the x4-unrolled step loop (no moves).  C and S of column xi+3 are fetched three steps ahead into a
// 4-slot register ring (the unrolled loop never moves the ring either).  (A skew of one column would
// serve H, V, Dg only: A's input would be produced in the same step.)
//
// Group 0 takes its V/Dg/A inputs from the band above through an "edge" buffer in HBM.  The last row of a
// band publishes its per-column state with 8-byte agent-scope (write-through) stores, drains them
// (s_waitcnt vmcnt(0)) and raises a per-chunk flag; in the band below an eighth "helper" wave polls the
// flag (relaxed, agent scope), fetches the records with agent-scope loads two batches ahead and parks
// them in an LDS ring, so the compute waves never wait on HBM for them (MI355X_MICROARCH.md: "8-B agent
// atomics both sides").  Bands depend only on the band above, and (pair, band) are handed out by an
// atomic ticket in arrival order, so a workgroup only ever waits for a workgroup that has already
// started: no co-residency requirement, no deadlock.  Every spin is bounded.
#pragma once

namespace camd {

// 7 compute waves + 1 helper wave = 8 waves per workgroup: two workgroups fill a CU's 16 wave slots at
// <= 128 VGPRs (nine waves would leave room for only one).
//   CAMD_BAND_MERGED 1 (measurement build; rounds 3 and 6): all 8 waves compute (bands of 8 * 64 / LANES rows) and wave
//       CAMD_BAND_DUTY_WAVE runs the helper's few instructions every CPB steps on the side, so that every SIMD carries
//       four compute waves instead of 4, 4, 3, 3.  Bit-exact, and no faster: first pass 15.07 against 15.05 ms per 64
//       pairs, MODE_SGBM last pass 11.4 / 11.6, MODE_HH last pass 23.0 against 19.1 (122 / 128 VGPRs, the latter with
//       100 bytes of scratch) -- profiles/r06_band_ab.txt.  The pass is not bound by its busiest SIMDs: half the
//       arithmetic (MODE_HH4's two directions) takes 12.4 ms, no S stores 12.9, both 8.5 (profiles/r06_band_probe.txt).
#ifndef CAMD_BAND_MERGED
#define CAMD_BAND_MERGED 0
#endif
#ifndef CAMD_BAND_COMPUTE_WAVES
#define CAMD_BAND_COMPUTE_WAVES (CAMD_BAND_MERGED ? 8 : 7)
#endif
#ifndef CAMD_BAND_HELPER_WAVE
#define CAMD_BAND_HELPER_WAVE CAMD_BAND_COMPUTE_WAVES    // !MERGED: which wave of the workgroup is the helper (default: the last)
#endif
// Cache policy of the volume accesses (bit 0: the C / S loads carry `nt`, bit 1: the S stores too).  Every volume byte is
// read once per pass; streamed past the L2's replacement order the first pass takes 15.1 instead of 15.3 ms and the
// row-parallel last pass 11.6 instead of 12.2 per 64 pairs (profiles/r06_band_nt.txt, three A/B repetitions).  `nt` on the
// S stores changes nothing (bit 1 off); `nt` on k_cost's 16-byte C stores defeats the L2's write combining: 72 ms.
#ifndef CAMD_BAND_NT
#define CAMD_BAND_NT 1
#endif
#ifndef CAMD_BAND_FLAG_DELAY
#define CAMD_BAND_FLAG_DELAY 0                           // steps between a chunk's last column and its flag (0 = drain at once)
#endif
#ifndef CAMD_ROW_PERSIST_PRIO
#define CAMD_ROW_PERSIST_PRIO 3                          // s_setprio of k_band_row_persist's waves (0 = leave it)
#endif
#ifndef CAMD_BAND_DUTY_WAVE
#define CAMD_BAND_DUTY_WAVE 0                            // MERGED: the compute wave that also feeds the edge ring
#endif
#ifndef CAMD_BAND_MIN_WAVES
#define CAMD_BAND_MIN_WAVES 4                            // occupancy target (waves per SIMD) of the D <= 128 instantiations
#endif
// the row-parallel (!FULL) pass has no helper duty: 1 = all eight waves of its workgroups compute (rows per workgroup =
// BAND_BLOCK / LANES), 0 = the wave that would be the helper just exits (round 1-4)
#ifndef CAMD_BAND_ROW_ALL_WAVES
#define CAMD_BAND_ROW_ALL_WAVES 0
#endif
#ifndef CAMD_BAND_ROW_MIN_WAVES
#define CAMD_BAND_ROW_MIN_WAVES CAMD_BAND_MIN_WAVES      // occupancy target of the row-parallel pass (it needs 66 VGPRs)
#endif
static constexpr int BAND_THREADS = 64 * CAMD_BAND_COMPUTE_WAVES;  // compute threads
static constexpr int BAND_BLOCK = BAND_THREADS + (CAMD_BAND_MERGED ? 0 : 64);  // (+ one helper wave)
static constexpr int BAND_HELPER_WAVE = CAMD_BAND_HELPER_WAVE;
static constexpr int BAND_RING = 4;                      // C/S prefetch ring (xi .. xi+3)
#ifndef BAND_RING_ROWS
#define BAND_RING_ROWS 4                                 // ... of the row-parallel (!FULL) pass
#endif
static constexpr int BAND_CHUNK = 16;                    // columns per edge flag
static constexpr uint32_t BAND_SPIN_LIMIT = 1u << 20;    // ~0.1 s of s_sleep polling, then give up

struct BandArgs {
    const uint16_t* C;
    uint16_t* S;
    unsigned long long* E;  // [pair][band] edge blocks: [W1][6*NQ][LANES] u64 vector words, then [W1][2] u64 deltas
    uint32_t* flags;        // [pair][band][nchunks], value == epoch when the chunk is published
    uint32_t* ticket;       // zeroed before every launch
    uint32_t* err;          // set to 1 if a bounded spin timed out
    uint32_t* keys;         // FINAL: right-view map keys [pair][H][W]
    int16_t* d1;            // FINAL: left disparity candidates [pair][H][W]
    size_t vol_stride;      // int16 elements per pair volume
    size_t erec_stride;     // u64 per (pair, band) edge block
    int sx, sy, nbands, nchunks, npairs;
    uint32_t epoch;
    int write_S;            // FINAL only: also store S (stage-wise parity hook)
};

// one SGM update: L = C + min(Lp, Lp[d-1]+P1, Lp[d+1]+P1, delta) - delta, packed u16, returns new delta
// edge_lo / edge_hi: two registers that live across all steps and directions.  A DPP row shift leaves the lane without
// a source (lane 0 of the row for row_shr, lane 15 for row_shl) untouched, so those lanes keep the MAX_COST sentinel
// they were initialised with and the other lanes are overwritten every time: no per-use reload of the sentinel.
template <int LANES, int NR, bool PAD>
__device__ __forceinline__ uint32_t sgm_step(const uint32_t (&Lp)[NR], uint32_t delta, const uint32_t (&c)[NR],
                                             uint32_t (&L)[NR], const uint32_t (&keep)[NR],
                                             const uint32_t (&sent)[NR], uint32_t P1pk, uint32_t P2pk, int li,
                                             uint32_t& edge_lo, uint32_t& edge_hi)
{
    edge_lo = dpp_mov<DPP_ROW_SHR1>(edge_lo, Lp[NR - 1]);
    edge_hi = dpp_mov<DPP_ROW_SHL1>(edge_hi, Lp[0]);
    uint32_t prev_last = edge_lo, next_first = edge_hi;
    if (LANES < 16) {
        if (li == 0) prev_last = SENT_PK;
        if (li == LANES - 1) next_first = SENT_PK;
    }
    uint32_t m[NR + 1];
    m[0] = alignbit16(Lp[0], prev_last);
#pragma unroll
    for (int k = 1; k < NR; k++) m[k] = alignbit16(Lp[k], Lp[k - 1]);
    m[NR] = alignbit16(next_first, Lp[NR - 1]);
    uint32_t mn = SENT_PK;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        // C + min(Lp, Lp[d-1] + P1, Lp[d+1] + P1, delta) - delta  ==  C - max(delta - min(Lp, ...), 0)
        // The + P1 and the final subtraction are plain 32-bit operations on the packed pair (v_add_u32 / v_sub_u32
        // issue at twice the rate of the v_pk_* forms on gfx950, tools/microtests/valu_rate.hip): inside the int16
        // regime no half can carry or borrow -- Lp <= 0x7fff and P1 < 0x8000, and L = C - (...) >= C - P2 >= 0
        // because delta - min(...) <= P2 <= C.  (Outside it -- a flagged volume -- the result is discarded anyway.)
        uint32_t t = pk_min_u16(m[k], m[k + 1]) + P1pk;
        uint32_t l = c[k] - pk_subsat_u16(delta, pk_min_u16(Lp[k], t));
        if (PAD) l = (l & keep[k]) | sent[k];  // d >= D carries MAX_COST
        L[k] = l;
        mn = pk_min_u16(mn, l);
    }
    return group_min_dup16<LANES>(mn) + P2pk;  // (no carry: min <= 0x7fff, P2 < 0x8000)
}


// u64 words of one (pair, band) edge block: per column LANES x 6NQ vector words, then 2 words of deltas
static inline size_t band_erec_stride(int W1, int lanes, int nq) { return (size_t)W1 * ((size_t)lanes * 6 * nq + 2); }  // nq = ceil(nr / 4)

// Winner-take-all on the final S of one pixel (bit-exact with k_wta), in two parts.
//   band_wta_step   every step, all lanes of the group: minS / best by a min-reduce over keys, the uniqueness
//                   test, and the two sub-pixel neighbours; the outcome is CAPTURED into lane (t mod LANES)
//   band_wta_flush  every LANES steps: each lane finishes the pixel it captured (right-view atomicMin,
//                   parabola, store) -- the scalar tail runs with all 64 lanes busy instead of one per group
// dpk[k] = the two disparities of register k, packed (d | d+1 << 16).
static constexpr uint32_t WTA_NONE = 0xffffffffu;
// The parked S vectors of the groups of a wave are read back at [best - 1] / [best + 1] with 2-byte reads (32-bank
// modulus, two groups per 32-lane half).  At a group stride of LANES * 16 bytes = a multiple of 128 the groups of a
// half hit the same bank whenever their winners share a dword -- neighbouring rows of a real scene nearly always do
// (round 4: SQ_LDS_BANK_CONFLICT = 24 % of the LDS-active cycles of the row-parallel pass).  CAMD_WTA_PADQ uint4 of padding
// per group (WtaPad) move the second group of a half by 16 banks: a conflict then needs winners 32 disparities apart.
#ifndef CAMD_WTA_PADQ
#define CAMD_WTA_PADQ 4
#endif
// (only where the group stride is a multiple of 128 bytes: 8 or 16 lanes per pixel; the slots of a lane are planes of
// their own -- lds_st_regs -- so the group stride does not depend on the registers per lane)
template <int LANES> struct WtaPad { static constexpr int Q = (LANES * 16) % 128 == 0 ? CAMD_WTA_PADQ : 0; };

// TIE8: MODE_SGBM_3WAY's winner among equal totals as OpenCV's CV_SIMD build picks it (k_wta, oracle way3_winner),
// for D % 8 == 0: disparities are scanned 8 at a time, each of the 8 lane slots keeps the LAST d attaining the
// minimum, the winner is the smallest of those positions.  With a single minimum that is the ordinary winner, so the
// rule is only evaluated when some pixel of the wave has its first and last minimum at different d.
template <int LANES, int NR, bool TIE8 = false, int NTH = BAND_THREADS>
__device__ __forceinline__ void band_wta_step(const uint32_t (&s)[NR], const uint32_t (&dpk)[NR], uint4* wS,
                                              const Geom& g, int ctid, int grp, int li, int t, bool act,
                                              uint32_t& cap_key, uint32_t& cap_nb)
{
    constexpr int NQ = (NR + 3) / 4;
    static_assert(!TIE8 || NR % 4 == 0, "the 8-slot tie rule needs whole groups of 8 disparities per lane");
    // (1) minS and the smallest d attaining it: min over keys (S << 16 | d); padded d >= D hold 0x7FFF
    uint32_t key = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        key = min(key, __builtin_amdgcn_perm(s[k], dpk[k], 0x05040100u));  // S.lo << 16 | d
        key = min(key, __builtin_amdgcn_perm(s[k], dpk[k], 0x07060302u));  // S.hi << 16 | d + 1
    }
    key = group_min_u32_full<LANES>(key);
    const int minS = (int)(key >> 16);
    int best = (int)(key & 0xffffu);
    if (TIE8) {
        uint32_t kl = 0xffffffffu;  // (S << 16 | 0xffff - d) minimum: smallest S, then LARGEST d
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const uint32_t nd = ~dpk[k];
            kl = min(kl, __builtin_amdgcn_perm(s[k], nd, 0x05040100u));
            kl = min(kl, __builtin_amdgcn_perm(s[k], nd, 0x07060302u));
        }
        kl = group_min_u32_full<LANES>(kl);
        if (__any((int)(0xffffu - (kl & 0xffffu)) != best)) {
            uint32_t pos = 0xffffffffu;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                uint32_t last = 0;  // 1 + the largest d of slot e (in this lane) whose total is the minimum
#pragma unroll
                for (int v = 0; v < NR / 4; v++) {
                    const uint32_t val = (e & 1) ? (s[4 * v + e / 2] >> 16) : (s[4 * v + e / 2] & 0xffffu);
                    const uint32_t d = (e & 1) ? (dpk[4 * v + e / 2] >> 16) : (dpk[4 * v + e / 2] & 0xffffu);
                    if ((int)d < g.D && (int)val == minS) last = d + 1;
                }
                last = group_max_u32<LANES>(last);
                if (last) pos = min(pos, last - 1);
            }
            best = (int)pos;
            key = ((uint32_t)minS << 16) | pos;
        }
    }
    // park S so that S[best-1], S[best+1] can be picked without a select tree
    // plane v of the parked vectors: [group][LANES + pad] uint4 (see WtaPad); WS_PLANE uint4 per plane
    constexpr int GST = LANES + WtaPad<LANES>::Q, WS_PLANE = (NTH / LANES) * GST;
    lds_st_regs<NR>(wS, grp * GST + li, WS_PLANE, s);
    // (2) uniqueness: not unique iff S[d]*(100-u) < minS*100 for some |d-best| > 1, i.e. iff that holds for the smallest
    //     such S[d] (100-u >= 1 on this path): two 24-bit multiplies of group-uniform values, no division
    //     (S <= 0xffff and 100-u <= 100: both products stay below 2^24)
    // the window best-1 .. best+1 is masked out in packed arithmetic: diff = d - (best-1) (mod 2^16) is 0, 1, 2
    // exactly there, w = sat(3 - diff) is non-zero exactly there, and sat(w * 0xFFFF + S) = 0xFFFF
    const uint32_t bm1 = dup16((uint32_t)(best - 1));
    uint32_t far = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const uint32_t w = pk_subsat_u16(0x00030003u, pk_sub_u16(dpk[k], bm1));
        far = pk_min_u16(far, pk_mad_sat_u16(w, 0xffffffffu, s[k]));
    }
    const int minfar = (int)(group_min_dup16<LANES>(far) & 0xffffu);
    const uint32_t uq = g.uniq <= 98 ? (uint32_t)(100 - g.uniq) : 1u;  // (uniq_magic == 0 stood for a divisor of 1)
    const bool ok = act && minS < MAX_COST && __umul24((uint32_t)minfar, uq) >= __umul24((uint32_t)minS, 100u);
    // (3) neighbours for the sub-pixel parabola (same address in every lane of the group: an LDS broadcast)
    const uint16_t* gs = reinterpret_cast<const uint16_t*>(wS + grp * GST);
    // disparity d sits in lane d / (2 NR), element w = d % (2 NR): plane w / 8, u16 (lane * 8 + w % 8) of the group's row
    auto slot_of = [](int d) {
        if (NQ == 1 && NR == 4) return d;
        const int ln = d / (2 * NR), w = d - ln * (2 * NR);
        return (w >> 3) * (WS_PLANE * 8) + ln * 8 + (w & 7);
    };
    const uint32_t Sm = gs[slot_of(max(best - 1, 0))], Sp = gs[slot_of(min(best + 1, LANES * 2 * NR - 1))];
    if (li == (t & (LANES - 1))) {
        cap_key = ok ? key : WTA_NONE;
        cap_nb = Sm | (Sp << 16);
    }
}

// x: column (cost coordinates) of the pixel this lane captured
template <int LANES>
__device__ __forceinline__ void band_wta_flush(const BandArgs& a, const Geom& g, int pair, int y, int x,
                                               uint32_t& cap_key, uint32_t cap_nb)
{
    if (cap_key != WTA_NONE) {
        const int minS = (int)(cap_key >> 16);
        int d = (int)(cap_key & 0xffffu);
        const int x2 = x + g.minX1 - d - g.minD;
        const size_t ro = ((size_t)pair * g.H + y) * (size_t)g.W;
        atomicMin(a.keys + ro + x2, ((uint32_t)minS << 16) | (uint32_t)(0xffff - d));
        if (0 < d && d < g.D - 1) {
            const int Sm = (int)(cap_nb & 0xffffu), Sp = (int)(cap_nb >> 16);
            const int denom2 = max(Sm + Sp - 2 * minS, 1);
            const int num = (Sm - Sp) * 16 + denom2, den = denom2 * 2;
            // num / den truncated toward zero; |quotient| <= 8: reciprocal estimate + fix-up
            int q = (int)((float)num * __frcp_rn((float)den));
            int r = num - q * den;
            if (num >= 0) {
                if (r < 0) q--;
                else if (r >= den) q++;
            } else {
                if (r > 0) q++;
                else if (r <= -den) q--;
            }
            d = d * 16 + q;
        } else
            d *= 16;
        a.d1[ro + x + g.minX1] = (int16_t)(d + g.minD * 16);
    }
    cap_key = WTA_NONE;
}

// FULL: H, V, Dg, A of sweep (sx, sy), skew 2.  !FULL: H of sweep sx only (rows independent).
// MODE: 0 = first pass (S written), 2 = final (S read, WTA; S stored only for the parity hook)
// DIAG = false (MODE_HH4): the two diagonal directions are left out (their slots travel as zeros)
// ALLW (row-parallel pass only): all waves of the workgroup compute (rows per ticket = BAND_BLOCK / LANES)
// ticket: the (pair, band) / run of rows this call works on (k_band: one per workgroup; k_band_row_persist: a loop)
template <int LANES, int NR, bool FULL, int MODE, bool PAD, bool DIAG, bool TIE8, bool ALLW>
__device__ __forceinline__ void band_body(const BandArgs& a, const Geom& g, const int ticket)
{
    static_assert(!(FULL && ALLW), "the wavefront pass keeps its helper wave");
    constexpr int NQ = (NR + 3) / 4;  // 16-byte LDS slots / u64 edge-record pairs per lane and vector
    constexpr int NTH = (!FULL && ALLW) ? BAND_BLOCK : BAND_THREADS;  // compute threads of a workgroup
    static_assert(!CAMD_BAND_MERGED || BAND_BLOCK == BAND_THREADS, "merged helper duty: every wave computes");
    constexpr int R = NTH / LANES;
    constexpr int EVEC = 6 * NQ;     // u64 per lane per column: V (2NQ), Dg (2NQ), A (2NQ)
    constexpr int CPB = 64 / LANES;  // columns the helper wave fetches per batch
    constexpr int RING = FULL ? BAND_RING : BAND_RING_ROWS;
    constexpr int SK = FULL ? 2 : 0;  // (staggering the rows of the row-parallel pass by 1 or 3 columns: no faster, profiles/r06_row_skew.txt)
    constexpr int XN = FULL ? BAND_THREADS * NQ : 1, EN = FULL ? 64 * NQ : 1;

    __shared__ uint4 xV[2][XN];
    __shared__ uint4 xD[2][XN];
    __shared__ uint4 xA[2][XN];
    __shared__ uint4 xdl[2][FULL ? R : 1];  // (deltaV, deltaDg, deltaA, -) per group
    __shared__ uint4 eV[3][EN];             // edge ring: 3 batches of CPB columns
    __shared__ uint4 eD[3][EN];
    __shared__ uint4 eA[3][EN];
    __shared__ uint4 edl[3][FULL ? CPB : 1];
    __shared__ uint4 wS[MODE == 2 ? NQ * R * (LANES + WtaPad<LANES>::Q) : 1];  // FINAL: S of the current pixel, per group

    // band-major order: band b of every pair is handed out before band b+1 of any pair, so with many
    // pairs in flight a workgroup's upstream band is usually far ahead by the time it starts (the
    // dependency (pair, b-1) always holds an earlier ticket)
    // (the row-parallel pass has no bands: its groups take consecutive rows of the whole batch -- see below)
    const int band = FULL ? ticket / a.npairs : 0;
    // roles: wave BAND_HELPER_WAVE is the helper, the others are compute waves numbered in wave order (ctid =
    // compute thread index).  Which wave helps decides which SIMD carries one compute wave less (see
    // tools/microtests/wave_simd_placement.hip)
    const int wv = threadIdx.x >> 6;
    const bool helper = !CAMD_BAND_MERGED && (FULL || !ALLW) && wv == BAND_HELPER_WAVE;  // wave-uniform
    if (!FULL && helper) return;
    const int ctid = (CAMD_BAND_MERGED || (!FULL && ALLW)) ? (int)threadIdx.x
                     : helper ? (int)(threadIdx.x & 63) : (((wv > BAND_HELPER_WAVE ? wv - 1 : wv) << 6) | (int)(threadIdx.x & 63));
    const int grp = ctid / LANES, li = ctid % LANES;
    const int W1 = g.W1, H = g.H;
    // FULL: row `grp` of band `band` of pair ticket % npairs.  !FULL: rows are independent, so the workgroups cut the
    // rows of ALL pairs into runs of R without regard to pair boundaries (no partly filled last band per pair: 480 rows
    // in bands of 56, 1080 in bands of 28); the groups past the last row of the batch idle on its last row
    const long long grow = FULL ? 0 : min((long long)ticket * R + grp, (long long)a.npairs * H - 1);
    const int pair = FULL ? ticket % a.npairs : (int)(grow / H);
    const int row = FULL ? band * R + grp : (int)(grow % H);  // row index in sweep order
    const bool rvalid = !helper && (FULL ? row < H : (long long)ticket * R + grp < (long long)a.npairs * H);
    const int y = a.sy > 0 ? row : H - 1 - row;
    const int glast = min(R, H - band * R) - 1;
    const bool has_prev = FULL && band > 0, has_next = FULL && band + 1 < a.nbands;
    const bool producer = !helper && has_next && grp == glast;
    // the wave that keeps the edge ring filled: the helper wave, or (MERGED) one compute wave on the side
    const bool hduty = FULL && has_prev && (CAMD_BAND_MERGED ? __builtin_amdgcn_readfirstlane(wv) == CAMD_BAND_DUTY_WAVE : helper);

    const uint32_t P1pk = dup16((uint32_t)g.P1), P2pk = dup16((uint32_t)g.P2);
    uint32_t keep[NR], sent[NR], dpk[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        int d0 = li * 2 * NR + 2 * k;
        dpk[k] = (uint32_t)d0 | ((uint32_t)(d0 + 1) << 16);
        uint32_t kp = (d0 < g.D ? 0xffffu : 0u) | (d0 + 1 < g.D ? 0xffff0000u : 0u);
        keep[k] = kp;
        sent[k] = ~kp & SENT_PK;
    }

    const size_t rowoff = (size_t)pair * a.vol_stride + ((size_t)(rvalid ? y : 0) * W1) * g.Dp + (size_t)li * (2 * NR);
    const uint16_t* Crow = a.C + rowoff;
    uint16_t* Srow = a.S + rowoff;
    auto cell_off = [&](int xi) -> size_t { return (size_t)(a.sx > 0 ? xi : W1 - 1 - xi) * g.Dp; };
    auto load_vec = [&](const uint16_t* p, uint32_t (&dst)[NR]) {
        if (CAMD_BAND_NT & 1) ld_regs_nt<NR>(p, dst);
        else ld_regs<NR>(p, dst);
    };
    auto lds_vec = [&](const uint4* p, int idx, int stride, uint32_t (&dst)[NR]) { lds_ld_regs<NR>(p, idx, stride, dst); };

    // ---- edge buffers ------------------------------------------------------------------------------
    unsigned long long* Eout = a.E + ((size_t)pair * a.nbands + band) * a.erec_stride;
    const unsigned long long* Ein = a.E + ((size_t)pair * a.nbands + (band > 0 ? band - 1 : 0)) * a.erec_stride;
    const size_t edelta = (size_t)W1 * LANES * EVEC;  // offset of the per-column delta words in an edge block
    uint32_t* Fout = a.flags + ((size_t)pair * a.nbands + band) * a.nchunks;
    const uint32_t* Fin = a.flags + ((size_t)pair * a.nbands + (band > 0 ? band - 1 : 0)) * a.nchunks;

    // helper wave state: lane hl fetches column (batch*CPB + hl/LANES), lane-in-group hl%LANES
    const int hl = threadIdx.x & 63;
    unsigned long long pend[EVEC], pdl[2];
#pragma unroll
    for (int k = 0; k < EVEC; k++) pend[k] = 0;
    pdl[0] = pdl[1] = 0;
    int edge_valid_upto = 0;  // columns [0, edge_valid_upto) of the band above are known to be published
    bool dead = false;        // a bounded wait expired: report it and stop waiting (results are then invalid)
    auto wait_cols = [&](int upto) {
        while (edge_valid_upto < upto) {
            const int chunk = edge_valid_upto / BAND_CHUNK;
            uint32_t spins = 0;
            while (!dead && __hip_atomic_load(Fin + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > BAND_SPIN_LIMIT) {
                    atomicOr(a.err, 1u);  // bit 0; bit 1 (a refused pair, k_poison_flagged) must survive
                    dead = true;
                }
            }
            edge_valid_upto = min((chunk + 1) * BAND_CHUNK, W1);
        }
    };
    auto fetch_batch = [&](int b) {
        const int col = b * CPB + hl / LANES;
        if (col < W1) {
            // record layout [column][word k][lane]: word k of the 16 lanes is one contiguous 128-byte run, so the
            // 8-byte write-through stores of a group (and these loads) fill whole 64-byte sectors
            const unsigned long long* p = Ein + (size_t)col * (LANES * EVEC) + (hl % LANES);
#pragma unroll
            for (int k = 0; k < EVEC; k++)
                pend[k] = __hip_atomic_load(p + k * LANES, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long* q = Ein + edelta + (size_t)col * 2;
            pdl[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pdl[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto park_batch = [&](int b) {
        const int slot = b % 3;
#pragma unroll
        for (int v = 0; v < NQ; v++) {
            eV[slot][v * 64 + hl] = make_uint4((uint32_t)pend[2 * v], (uint32_t)(pend[2 * v] >> 32),
                                               (uint32_t)pend[2 * v + 1], (uint32_t)(pend[2 * v + 1] >> 32));
            eD[slot][v * 64 + hl] =
                make_uint4((uint32_t)pend[2 * NQ + 2 * v], (uint32_t)(pend[2 * NQ + 2 * v] >> 32),
                           (uint32_t)pend[2 * NQ + 2 * v + 1], (uint32_t)(pend[2 * NQ + 2 * v + 1] >> 32));
            eA[slot][v * 64 + hl] =
                make_uint4((uint32_t)pend[4 * NQ + 2 * v], (uint32_t)(pend[4 * NQ + 2 * v] >> 32),
                           (uint32_t)pend[4 * NQ + 2 * v + 1], (uint32_t)(pend[4 * NQ + 2 * v + 1] >> 32));
        }
        if (hl % LANES == 0)
            edl[slot][hl / LANES] = make_uint4((uint32_t)pdl[0], (uint32_t)(pdl[0] >> 32), (uint32_t)pdl[1], 0u);
    };
    if (hduty) {
        wait_cols(min(CPB, W1));
        fetch_batch(0);
        park_batch(0);
        if (CPB < W1) {
            wait_cols(min(2 * CPB, W1));
            fetch_batch(1);
        }
    }

    // ---- compute state -----------------------------------------------------------------------------
    uint32_t LH[NR];
    uint32_t VS[2][NR], DS[4][NR], dVs[2], dDs[4];  // V / Dg inputs in flight (rotating with the unrolled loop)
#pragma unroll
    for (int k = 0; k < NR; k++) {
        LH[k] = 0;
        VS[0][k] = VS[1][k] = 0;
        DS[0][k] = DS[1][k] = DS[2][k] = DS[3][k] = 0;
    }
    uint32_t dH = P2pk;
    uint32_t edge_lo = SENT_PK, edge_hi = SENT_PK;  // see sgm_step
    dVs[0] = dVs[1] = P2pk;
    dDs[0] = dDs[1] = dDs[2] = dDs[3] = P2pk;

    uint32_t cr[RING][NR], sr[RING][NR];
#pragma unroll
    for (int u = 0; u < RING; u++)
#pragma unroll
        for (int k = 0; k < NR; k++) { cr[u][k] = 0; sr[u][k] = 0; }
#pragma unroll
    for (int u = 0; u < RING - 1; u++) {
        const int xp = min(max(u - SK * grp, 0), W1 - 1);
        if (!helper) {
            load_vec(Crow + cell_off(xp), cr[u]);
            if (MODE != 0) load_vec(Srow + cell_off(xp), sr[u]);
        }
    }
    if (FULL) {
        if (!helper) {
            // slot 1 is what step 0 reads as "produced in step -1": the zero border state
#pragma unroll
            for (int v = 0; v < NQ; v++) {
                xV[1][v * BAND_THREADS + ctid] = make_uint4(0u, 0u, 0u, 0u);
                xD[1][v * BAND_THREADS + ctid] = make_uint4(0u, 0u, 0u, 0u);
                xA[1][v * BAND_THREADS + ctid] = make_uint4(0u, 0u, 0u, 0u);
            }
            if (li == 0) xdl[1][grp] = make_uint4(P2pk, P2pk, P2pk, 0u);
        }
        __syncthreads();  // edge batch 0 is parked
        if (!helper && grp == 0 && has_prev) {
            // column 0 of the band above: V input of step 0 (set 1), Dg input of step 1 (set 3)
            lds_vec(eV[0], li, 64, VS[1]);
            lds_vec(eD[0], li, 64, DS[3]);
            const uint4 dl = edl[0][0];
            dVs[1] = dl.x;
            dDs[3] = dl.y;
        }
    }

    const int nsteps = (W1 + SK * glast + RING - 1) / RING * RING;
    uint32_t cap_key = WTA_NONE, cap_nb = 0;  // FINAL: the pixel this lane finishes at the next flush
    // every CPB steps: batch t / CPB + 1 (fetched CPB steps ago) goes into the LDS ring, the batch after it is requested
    auto duty_step = [&](int t) {
        const int b = t / CPB + 1;
        if (b * CPB < W1) park_batch(b);
        const int bn = b + 1;
        if (bn * CPB < W1) {
            wait_cols(min((bn + 1) * CPB, W1));
            fetch_batch(bn);
        }
    };
    if (FULL && helper) {
        // The helper runs its own loop with the same number of barriers: keeping the two roles in separate
        // loops leaves the compute loop free of control flow around its loads, which is what lets the
        // compiler count them (s_waitcnt vmcnt(N), N > 0) instead of draining the prefetch ring every step.
        for (int t = 0; t < nsteps; t++) {
            if (has_prev && (t % CPB) == 0) duty_step(t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef CAMD_BAND_DBG_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
        }
        return;
    }
    for (int t0 = 0; t0 < nsteps; t0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; u++) {
            const int t = t0 + u;
            const int xi = t - SK * grp;
            const bool act = rvalid && xi >= 0 && xi < W1;
            if (CAMD_BAND_MERGED && FULL) {
                // (wave-uniform; with CPB == RING a compile-time position in the unrolled body)
                if ((CPB == RING ? u == 0 : (t % CPB) == 0) && hduty) duty_step(t);
            }
            {
                constexpr int UP = (RING - 1);
                const int xp = min(max(xi + UP, 0), W1 - 1);
                load_vec(Crow + cell_off(xp), cr[(u + UP) % RING]);
                if (MODE != 0) load_vec(Srow + cell_off(xp), sr[(u + UP) % RING]);
            }
            const uint32_t(&cc)[NR] = cr[u];
            // ---- the row above's vectors of column xi+1 (produced in the previous step)
            uint32_t An[NR], dAn = P2pk;
#pragma unroll
            for (int k = 0; k < NR; k++) An[k] = 0;
            if (FULL) {
                const int rb = (u & 1) ^ 1;
                if (grp > 0) {
                    lds_vec(xV[rb], ctid - LANES, BAND_THREADS, VS[u & 1]);
                    lds_vec(xD[rb], ctid - LANES, BAND_THREADS, DS[u & 3]);
                    lds_vec(xA[rb], ctid - LANES, BAND_THREADS, An);
                    const uint4 dl = xdl[rb][grp - 1];
                    dVs[u & 1] = dl.x;
                    dDs[u & 3] = dl.y;
                    dAn = dl.z;
                } else if (has_prev && xi + 1 < W1) {
                    const int col = xi + 1;
                    const int slot = (col / CPB) % 3, e = (col % CPB) * LANES + li;
                    lds_vec(eV[slot], e, 64, VS[u & 1]);
                    lds_vec(eD[slot], e, 64, DS[u & 3]);
                    lds_vec(eA[slot], e, 64, An);
                    const uint4 dl = edl[slot][col % CPB];
                    dVs[u & 1] = dl.x;
                    dDs[u & 3] = dl.y;
                    dAn = dl.z;
                }
            }
            const uint32_t(&Vin)[NR] = VS[(u + 1) & 1];
            const uint32_t(&Din)[NR] = DS[(u + 2) & 3];
            const uint32_t dV = dVs[(u + 1) & 1], dD = dDs[(u + 2) & 3];

            uint32_t LVo[NR], LDo[NR], LAo[NR], dVo = P2pk, dDo = P2pk, dAo = P2pk;
#pragma unroll
            for (int k = 0; k < NR; k++) { LVo[k] = 0; LDo[k] = 0; LAo[k] = 0; }
            if (act) {
                uint32_t s[NR];
#pragma unroll
                for (int k = 0; k < NR; k++) s[k] = MODE != 0 ? sr[u][k] : 0u;
                {
                    uint32_t L[NR];
                    dH = sgm_step<LANES, NR, PAD>(LH, dH, cc, L, keep, sent, P1pk, P2pk, li, edge_lo, edge_hi);
#pragma unroll
                    for (int k = 0; k < NR; k++) { LH[k] = L[k]; s[k] = pk_addsat_i16(s[k], L[k]); }
                }
                if (FULL) {
                    dVo = sgm_step<LANES, NR, PAD>(Vin, dV, cc, LVo, keep, sent, P1pk, P2pk, li, edge_lo, edge_hi);
#pragma unroll
                    for (int k = 0; k < NR; k++) s[k] = pk_addsat_i16(s[k], LVo[k]);
                    if (DIAG) {
                        dDo = sgm_step<LANES, NR, PAD>(Din, dD, cc, LDo, keep, sent, P1pk, P2pk, li, edge_lo, edge_hi);
#pragma unroll
                        for (int k = 0; k < NR; k++) s[k] = pk_addsat_i16(s[k], LDo[k]);
                        dAo = sgm_step<LANES, NR, PAD>(An, dAn, cc, LAo, keep, sent, P1pk, P2pk, li, edge_lo, edge_hi);
#pragma unroll
                        for (int k = 0; k < NR; k++) s[k] = pk_addsat_i16(s[k], LAo[k]);
                    }
                }
#if defined(CAMD_BAND_DBG_NOSTORE) && !defined(CAMD_MEASUREMENT_BUILD)
#error "CAMD_BAND_DBG_NOSTORE produces wrong results: measurement builds only (define CAMD_MEASUREMENT_BUILD too)"
#endif
#ifdef CAMD_BAND_DBG_NOSTORE  // (measurement only: the pass without its V of stores; S must still look used)
                if (s[0] == 0x12345678u) st_regs<NR>(Srow + cell_off(xi), s);
#else
                if (MODE != 2 || a.write_S) {
                    if (CAMD_BAND_NT & 2) st_regs_nt<NR>(Srow + cell_off(xi), s);
                    else st_regs<NR>(Srow + cell_off(xi), s);
                }
#endif
                if (MODE == 2) band_wta_step<LANES, NR, TIE8, NTH>(s, dpk, wS, g, ctid, grp, li, t, true, cap_key, cap_nb);
            }
            if (MODE == 2 && ((t & (LANES - 1)) == LANES - 1 || t == nsteps - 1)) {
                // lane li captured the pixel of step t - ((t mod LANES) - li)
                const int xc = xi - ((t & (LANES - 1)) - li);
                band_wta_flush<LANES>(a, g, pair, y, a.sx > 0 ? xc : W1 - 1 - xc, cap_key, cap_nb);
            }
            if (FULL) {
                const int wb = u & 1;
                lds_st_regs<NR>(xV[wb], ctid, BAND_THREADS, LVo);
                lds_st_regs<NR>(xD[wb], ctid, BAND_THREADS, LDo);
                lds_st_regs<NR>(xA[wb], ctid, BAND_THREADS, LAo);
                if (li == 0) xdl[wb][grp] = make_uint4(dVo, dDo, dAo, 0u);
                if (producer && act) {
                    unsigned long long* p = Eout + (size_t)xi * (LANES * EVEC) + li;
#pragma unroll
                    for (int k = 0; k < 2 * NQ; k++) {
                        if (2 * k >= NR) break;  // (slots past the lane's registers are never read back as anything but padding)
                        __hip_atomic_store(p + k * LANES, (unsigned long long)LVo[2 * k] | ((unsigned long long)reg_or0<NR>(LVo, 2 * k + 1) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p + (2 * NQ + k) * LANES,
                                           (unsigned long long)LDo[2 * k] | ((unsigned long long)reg_or0<NR>(LDo, 2 * k + 1) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p + (4 * NQ + k) * LANES,
                                           (unsigned long long)LAo[2 * k] | ((unsigned long long)reg_or0<NR>(LAo, 2 * k + 1) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (li == 0) {
                        unsigned long long* q = Eout + edelta + (size_t)xi * 2;
                        __hip_atomic_store(q, (unsigned long long)dVo | ((unsigned long long)dDo << 32), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(q + 1, (unsigned long long)dAo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#if CAMD_BAND_FLAG_DELAY == 0
                    if ((xi % BAND_CHUNK) == BAND_CHUNK - 1 || xi == W1 - 1) {
                        // the write-through stores of the whole chunk must have landed before the flag is raised
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (li == 0)
                            __hip_atomic_store(Fout + xi / BAND_CHUNK, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#else
                    // The write-through stores of a whole chunk must have landed before its flag is raised.  Waiting for
                    // them right behind the chunk's last store (s_waitcnt vmcnt(0)) also drains this wave's C / S prefetch
                    // ring and the stores just issued: one full memory latency every BAND_CHUNK steps during which the
                    // other six waves stand at the barrier.  The memory counter retires in order, so the flag of a chunk is
                    // raised FLAG_DELAY steps after its last column instead, behind a COUNTED wait: every step the producer
                    // group is active in issues at least OPS vector-memory operations in this wave (the C load, the S store
                    // or load, the edge-record stores), so once all but the newest FLAG_DELAY * OPS have retired, everything
                    // up to the chunk's last store has.  The last column of the row drains and raises what is left.
                    constexpr int FLAG_DELAY = CAMD_BAND_FLAG_DELAY;
                    constexpr int OPS = 3 * ((NR + 1) / 2) + 2 + 2;  // (lower bound: edge vectors + 2 delta stores + C + S)
                    static_assert(FLAG_DELAY * OPS <= 63, "vmcnt is a 6-bit counter");
                    if (xi == W1 - 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        // chunks whose delayed raise (at column 16 m + 15 + FLAG_DELAY) lies before this column are up
                        const int up = W1 - 2 - FLAG_DELAY - (BAND_CHUNK - 1) >= 0 ? (W1 - 2 - FLAG_DELAY - (BAND_CHUNK - 1)) / BAND_CHUNK + 1 : 0;
                        if (li < a.nchunks - up)
                            __hip_atomic_store(Fout + up + li, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (xi >= FLAG_DELAY && ((xi - FLAG_DELAY) % BAND_CHUNK) == BAND_CHUNK - 1) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FLAG_DELAY * OPS) : "memory");
                        if (li == 0)
                            __hip_atomic_store(Fout + (xi - FLAG_DELAY) / BAND_CHUNK, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#endif
                }
                // LDS-only barrier (a __syncthreads() would drain the C/S prefetch ring)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(CAMD_BAND_DBG_NOBARRIER) && !defined(CAMD_MEASUREMENT_BUILD)
#error "CAMD_BAND_DBG_NOBARRIER produces wrong results: measurement builds only (define CAMD_MEASUREMENT_BUILD too)"
#endif
#ifndef CAMD_BAND_DBG_NOBARRIER  // (measurement only: what the per-step workgroup barrier costs)
                __builtin_amdgcn_s_barrier();
#endif
            }
        }
    }
}

template <int LANES, int NR, bool FULL, int MODE, bool PAD, bool DIAG = true, bool TIE8 = false>
__global__ __launch_bounds__(BAND_BLOCK, NR <= 4 ? (FULL ? CAMD_BAND_MIN_WAVES : CAMD_BAND_ROW_MIN_WAVES) : 2) void k_band(BandArgs a, Geom g)
{
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.ticket, 1u);
    __syncthreads();
    band_body<LANES, NR, FULL, MODE, PAD, DIAG, TIE8, !FULL && CAMD_BAND_ROW_ALL_WAVES>(a, g, (int)s_ticket);
}

// The row-parallel last pass with a FIXED number of resident workgroups (grid = workgroups per CU x 256) that take runs
// of BAND_BLOCK / LANES rows by ticket until the batch is done: the counterpart of k_cost_persist (sgbm_cost.hpp) --
// launched this way the pass leaves room on every CU for the cost kernel of the next batch.
template <int LANES, int NR, bool PAD, bool TIE8 = false>
__global__ __launch_bounds__(BAND_BLOCK, NR <= 4 ? CAMD_BAND_ROW_MIN_WAVES : 2) void k_band_row_persist(BandArgs a, Geom g)
{
    __shared__ uint32_t s_ticket;
    const long long rows = (long long)a.npairs * g.H;
    // Beside the cost kernel's waves -- older, and ready to issue VALU every cycle -- these waves would get the issue
    // slots that are left over ("priority, then age"), i.e. next to none, and the pass would crawl until the cost kernel
    // is done.  Raised priority gives the pass the ~half of the VALU issue it needs to keep HBM busy; the cost kernel
    // takes the rest.
#if CAMD_ROW_PERSIST_PRIO
    __builtin_amdgcn_s_setprio(CAMD_ROW_PERSIST_PRIO);
#endif
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const int ticket = (int)s_ticket;
        __syncthreads();
        if ((long long)ticket * (BAND_BLOCK / LANES) >= rows) break;
        band_body<LANES, NR, false, 2, PAD, true, TIE8, true>(a, g, ticket);
    }
}

// left-right check of one row from the candidates / right-view keys the FINAL band pass produced; a thread owns the
// pixel pair (2i, 2i+1): candidates in and disparities out travel as one dword per pair
constexpr int LRCHECK_ROWS = 8;  // rows a block walks (one row per block was bound by the rate at which workgroups start)
__global__ __launch_bounds__(256) void k_lrcheck(const int16_t* __restrict__ d1, const uint32_t* __restrict__ keys,
                                                 int16_t* __restrict__ out, size_t out_pitch_e, size_t out_stride_e,
                                                 Geom g, const uint32_t* __restrict__ err)
{
    const int x0 = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (x0 >= g.W) return;
    const bool two = x0 + 1 < g.W;
    const int ya = blockIdx.y * LRCHECK_ROWS, yb = min(ya + LRCHECK_ROWS, g.H), pair = blockIdx.z;
    const int INVALID_SCALED = (g.minD - 1) * 16;
    // a band whose bounded wait expired computed on unpublished edge data: poison the whole result rather
    // than hand back plausible-looking garbage (camd_sgbm_status / the next compute report the error)
    const bool poisoned = (*err & 1u) != 0;  // bit 0 = a band pass timed out (bit 1, a refused pair, is per pair: k_poison_flagged)
    uint32_t both[LRCHECK_ROWS];
#pragma unroll
    for (int k = 0; k < LRCHECK_ROWS; k++) {  // all candidate rows requested before the first dependent gather
        const size_t ro = ((size_t)pair * g.H + min(ya + k, g.H - 1)) * (size_t)g.W;
        if (two) __builtin_memcpy(&both[k], d1 + ro + x0, 4);
        else both[k] = (uint16_t)d1[ro + x0];
    }
#pragma unroll
    for (int k = 0; k < LRCHECK_ROWS; k++) {
        const int y = ya + k;
        if (y >= yb) break;
        const size_t ro = ((size_t)pair * g.H + y) * (size_t)g.W;
        auto check = [&](int x, int v1) -> int {
            if (x >= g.minX1 && x < g.minX1 + g.W1 && v1 != INVALID_SCALED) {
                int _d = v1 >> 4, d_ = (v1 + 15) >> 4;
                int _x = x - _d, x_ = x - d_;
                bool bad = false;
                if (0 <= _x && _x < g.W) {
                    uint32_t kk = keys[ro + _x];
                    int v = kk == KEY_INIT ? INVALID_SCALED : (int)(0xffffu - (kk & 0xffffu)) + g.minD;
                    bad = v >= g.minD && abs(v - _d) > g.d12;
                }
                if (bad) {
                    bad = false;
                    if (0 <= x_ && x_ < g.W) {
                        uint32_t kk = keys[ro + x_];
                        int v = kk == KEY_INIT ? INVALID_SCALED : (int)(0xffffu - (kk & 0xffffu)) + g.minD;
                        bad = v >= g.minD && abs(v - d_) > g.d12;
                    }
                }
                if (bad) v1 = INVALID_SCALED;
            }
            return poisoned ? INVALID_SCALED : v1;
        };
        const uint32_t lo = (uint32_t)check(x0, (int)(int16_t)(both[k] & 0xffffu)) & 0xffffu;
        int16_t* o = out + (size_t)pair * out_stride_e + (size_t)y * out_pitch_e + x0;
        if (two) {
            const uint32_t r = lo | ((uint32_t)check(x0 + 1, (int)(int16_t)(both[k] >> 16)) << 16);
            __builtin_memcpy(o, &r, 4);
        } else
            *o = (int16_t)lo;
    }
}

__global__ __launch_bounds__(256) void k_wta_init(uint32_t* keys, int16_t* d1, size_t n, int invalid)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        keys[i] = KEY_INIT;
        d1[i] = (int16_t)invalid;
    }
}

}  // namespace camd

// sgbm_band.hpp -- fused multi-direction aggregation for gfx950: the "band wavefront" pass.
// Included by sgbm.hip (shares Geom, SENT_PK, KEY_INIT).
//
// One pass aggregates up to three directions that share a sweep orientation (sx, sy):
//     H = (sx, 0)      V = (0, sy)      Dg = (sx, sy)
// so that the cost volume C is read once and S is read-modified-written once for all of them
// (3V of HBM traffic per pass instead of 3V per direction).  MODE_SGBM = 2 passes
// {->, v, \} then {<-, /}+WTA;  MODE_HH = 4 passes {->, v, \}, {<-, ^, \^}, {/}, {/^}+WTA.
//
// Work decomposition: the image is cut into bands of R = BAND_THREADS/LANES rows.  A workgroup owns a
// band; its group g (LANES lanes, one pixel's disparity vector) owns row g of the band and walks it in
// sweep order.  At step t group g is at column index xi = t - g (a skewed wavefront), therefore
//     H  input  (xi-1, row)    = the group's own registers,
//     V  input  (xi,   row-1)  = what group g-1 produced in step t-1,
//     Dg input  (xi-1, row-1)  = what group g-1 produced in step t-2 (fetched at t-1, held one step),
// exchanged through a double-buffered LDS slot per group with ONE barrier per step.  C and S of column
// xi+3 are fetched three steps ahead into a 4-slot register ring (the step loop is unrolled by 4 so the
// ring never moves).
//
// Group 0 takes its V/Dg inputs from the band above through an "edge" buffer in HBM.  The last row of a
// band publishes its per-column state with 8-byte agent-scope (write-through) stores, drains them
// (s_waitcnt vmcnt(0)) and raises a per-chunk flag; in the band below a ninth "helper" wave polls the
// flag (relaxed, agent scope), fetches the records with agent-scope loads two batches ahead and parks
// them in an LDS ring, so the compute waves never wait on HBM for them (MI355X_MICROARCH.md: "8-B agent
// atomics both sides").  Bands depend only on the band above, and (pair, band) are handed out by an
// atomic ticket in arrival order, so a workgroup only ever waits for a workgroup that has already
// started: no co-residency requirement, no deadlock.  Every spin is bounded.
#pragma once

namespace camd {

// 7 compute waves + 1 helper wave = 8 waves per workgroup: two workgroups fill a CU's 16 wave slots at
// <= 128 VGPRs (nine waves would leave room for only one)
static constexpr int BAND_THREADS = 448;                 // compute threads
static constexpr int BAND_BLOCK = BAND_THREADS + 64;     // + one helper wave
static constexpr int BAND_RING = 4;                      // C/S prefetch ring (xi .. xi+3)
static constexpr int BAND_CHUNK = 16;                    // columns per edge flag
static constexpr uint32_t BAND_SPIN_LIMIT = 1u << 20;    // ~0.1 s of s_sleep polling, then give up

struct BandArgs {
    const uint16_t* C;
    uint16_t* S;
    unsigned long long* E;  // [pair][band][W1][LANES][4*NV+1] u64 edge records
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
    int nodep;              // measurement only (CAMD_BAND_NODEP=1): cut the band->band dependency (wrong results)
};

// one SGM update: L = C + min(Lp, Lp[d-1]+P1, Lp[d+1]+P1, delta) - delta, packed u16, returns new delta
template <int LANES, int NR, bool PAD>
__device__ __forceinline__ uint32_t sgm_step(const uint32_t (&Lp)[NR], uint32_t delta, const uint32_t (&c)[NR],
                                             uint32_t (&L)[NR], const uint32_t (&keep)[NR],
                                             const uint32_t (&sent)[NR], uint32_t P1pk, uint32_t P2pk, int li)
{
    uint32_t prev_last = dpp_mov<DPP_ROW_SHR1>(SENT_PK, Lp[NR - 1]);
    uint32_t next_first = dpp_mov<DPP_ROW_SHL1>(SENT_PK, Lp[0]);
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
        uint32_t t = pk_add_u16(pk_min_u16(m[k], m[k + 1]), P1pk);
        uint32_t a = pk_min_u16(pk_min_u16(Lp[k], t), delta);
        uint32_t l = pk_sub_u16(pk_add_u16(c[k], a), delta);
        if (PAD) l = (l & keep[k]) | sent[k];  // d >= D carries MAX_COST
        L[k] = l;
        mn = pk_min_u16(mn, l);
    }
    return pk_add_u16(group_min_dup16<LANES>(mn), P2pk);
}

// DIRS: bit0 H, bit1 V, bit2 Dg.  MODE: 0 = first pass (S written), 1 = middle (S += ...), 2 = final (WTA)
// PAD: Dp > D (padded disparities are forced to MAX_COST after every update)
template <int LANES, int NV, int DIRS, int MODE, bool PAD>
__global__ __launch_bounds__(BAND_BLOCK) void k_band(BandArgs a, Geom g)
{
    constexpr int NR = 4 * NV;
    constexpr int R = BAND_THREADS / LANES;
    constexpr bool HAS_H = (DIRS & 1) != 0, HAS_V = (DIRS & 2) != 0, HAS_D = (DIRS & 4) != 0;
    constexpr int EREC = 4 * NV + 1;   // u64 per lane per column: V (2NV), Dg (2NV), deltas (1)
    constexpr int CPB = 64 / LANES;    // columns the helper wave fetches per batch
    constexpr int RING = BAND_RING;

    __shared__ uint4 xV[2][BAND_THREADS * NV];
    __shared__ uint4 xD[2][BAND_THREADS * NV];
    __shared__ uint32_t xdV[2][R];
    __shared__ uint32_t xdD[2][R];
    __shared__ uint4 eVl[3][64 * NV];  // edge ring: 3 batches of CPB columns
    __shared__ uint4 eDl[3][64 * NV];
    __shared__ uint32_t edVl[3][CPB];
    __shared__ uint32_t edDl[3][CPB];
    __shared__ uint4 wS[MODE == 2 ? BAND_THREADS * NV : 1];  // FINAL: S of the current pixel, per group
    __shared__ uint32_t s_ticket;

    if (threadIdx.x == 0) s_ticket = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const int ticket = (int)s_ticket;
    // band-major order: band b of every pair is handed out before band b+1 of any pair, so with many
    // pairs in flight a workgroup's upstream band is usually far ahead by the time it starts (the
    // dependency (pair, b-1) always holds an earlier ticket)
    const int band = ticket / a.npairs, pair = ticket % a.npairs;
    const bool helper = threadIdx.x >= BAND_THREADS;  // wave-uniform
    const int grp = threadIdx.x / LANES, li = threadIdx.x % LANES;
    const int W1 = g.W1, H = g.H;
    const int row = band * R + grp;  // row index in sweep order
    const bool rvalid = !helper && row < H;
    const int y = a.sy > 0 ? row : H - 1 - row;
    const int glast = min(R, H - band * R) - 1;
    const bool has_prev = band > 0 && !a.nodep, has_next = band + 1 < a.nbands && !a.nodep;
    const bool producer = !helper && has_next && grp == glast;

    const uint32_t P1pk = dup16((uint32_t)g.P1), P2pk = dup16((uint32_t)g.P2);
    uint32_t keep[NR], sent[NR], dlo[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        int d0 = li * 8 * NV + 2 * k;
        dlo[k] = (uint32_t)d0;
        uint32_t kp = (d0 < g.D ? 0xffffu : 0u) | (d0 + 1 < g.D ? 0xffff0000u : 0u);
        keep[k] = kp;
        sent[k] = ~kp & SENT_PK;
    }

    // row base (element offsets); cell xi lives at column x = sx > 0 ? xi : W1-1-xi
    const size_t rowoff = (size_t)pair * a.vol_stride + ((size_t)(rvalid ? y : 0) * W1) * g.Dp + (size_t)li * (8 * NV);
    const uint16_t* Crow = a.C + rowoff;
    uint16_t* Srow = a.S + rowoff;
    auto cell_off = [&](int xi) -> size_t { return (size_t)(a.sx > 0 ? xi : W1 - 1 - xi) * g.Dp; };
    auto load_vec = [&](const uint16_t* p, uint32_t (&dst)[NR]) {
        const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            uint4 w = q[v];
            dst[4 * v] = w.x; dst[4 * v + 1] = w.y; dst[4 * v + 2] = w.z; dst[4 * v + 3] = w.w;
        }
    };

    // ---- edge buffers ------------------------------------------------------------------------------
    unsigned long long* Eout = a.E + ((size_t)pair * a.nbands + band) * a.erec_stride;
    const unsigned long long* Ein = a.E + ((size_t)pair * a.nbands + (band > 0 ? band - 1 : 0)) * a.erec_stride;
    uint32_t* Fout = a.flags + ((size_t)pair * a.nbands + band) * a.nchunks;
    const uint32_t* Fin = a.flags + ((size_t)pair * a.nbands + (band > 0 ? band - 1 : 0)) * a.nchunks;

    // helper wave state: lane hl fetches column (batch*CPB + hl/LANES), lane-in-group hl%LANES
    const int hl = threadIdx.x - BAND_THREADS;
    unsigned long long pend[EREC];
#pragma unroll
    for (int k = 0; k < EREC; k++) pend[k] = 0;
    int edge_valid_upto = 0;  // columns [0, edge_valid_upto) of the band above are known to be published
    bool dead = false;        // a bounded wait expired: report it and stop waiting (results are then invalid)
    auto wait_cols = [&](int upto) {  // block until columns [0, upto) of the band above are published
        while (edge_valid_upto < upto) {
            const int chunk = edge_valid_upto / BAND_CHUNK;
            uint32_t spins = 0;
            while (!dead && __hip_atomic_load(Fin + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > BAND_SPIN_LIMIT) {
                    atomicExch(a.err, 1u);
                    dead = true;
                }
            }
            edge_valid_upto = min((chunk + 1) * BAND_CHUNK, W1);
        }
    };
    auto fetch_batch = [&](int b) {  // issue the loads of batch b into pend
        const int col = b * CPB + hl / LANES;
        if (col < W1) {
            const unsigned long long* p = Ein + ((size_t)col * LANES + (hl % LANES)) * EREC;
#pragma unroll
            for (int k = 0; k < EREC; k++) pend[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto park_batch = [&](int b) {  // pend -> LDS ring slot b % 3
        const int slot = b % 3;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            eVl[slot][hl * NV + v] = make_uint4((uint32_t)pend[2 * v], (uint32_t)(pend[2 * v] >> 32),
                                               (uint32_t)pend[2 * v + 1], (uint32_t)(pend[2 * v + 1] >> 32));
            eDl[slot][hl * NV + v] =
                make_uint4((uint32_t)pend[2 * NV + 2 * v], (uint32_t)(pend[2 * NV + 2 * v] >> 32),
                           (uint32_t)pend[2 * NV + 2 * v + 1], (uint32_t)(pend[2 * NV + 2 * v + 1] >> 32));
        }
        if (hl % LANES == 0) {
            edVl[slot][hl / LANES] = (uint32_t)pend[4 * NV];
            edDl[slot][hl / LANES] = (uint32_t)(pend[4 * NV] >> 32);
        }
    };
    if (helper && has_prev) {
        wait_cols(min(CPB, W1));
        fetch_batch(0);
        park_batch(0);
        if (CPB < W1) {
            wait_cols(min(2 * CPB, W1));
            fetch_batch(1);
        }
    }

    // ---- compute state -----------------------------------------------------------------------------
    uint32_t LH[NR], Dh[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) { LH[k] = 0; Dh[k] = 0; }
    uint32_t dH = P2pk, dDh = P2pk;

    // prefetch ring: slot (t % RING) holds column xi of step t; columns xi .. xi+RING-2 are in flight
    uint32_t cr[RING][NR], sr[RING][NR];
#pragma unroll
    for (int u = 0; u < RING; u++)
#pragma unroll
        for (int k = 0; k < NR; k++) { cr[u][k] = 0; sr[u][k] = 0; }
#pragma unroll
    for (int u = 0; u < RING - 1; u++) {
        const int xp = min(max(u - grp, 0), W1 - 1);
        if (!helper) {
            load_vec(Crow + cell_off(xp), cr[u]);
            if (MODE != 0) load_vec(Srow + cell_off(xp), sr[u]);
        }
    }
    __syncthreads();  // edge batch 0 is parked

    // Steps are padded to a multiple of RING (padding steps have no active cell).  The helper wave runs
    // its own loop with the same number of barriers: keeping the two roles in separate loops leaves the
    // compute loop free of control flow around its loads, which is what lets the compiler count them
    // (s_waitcnt vmcnt(N), N > 0) instead of draining the prefetch ring every step.
    const int nsteps = (W1 + glast + RING - 1) / RING * RING;
    if (helper) {
        for (int t = 0; t < nsteps; t++) {
            if (has_prev && (t % CPB) == 0) {
                // park the batch fetched CPB steps ago, fetch the one after it
                const int b = t / CPB + 1;
                if (b * CPB < W1) park_batch(b);
                const int bn = b + 1;
                if (bn * CPB < W1) {
                    wait_cols(min((bn + 1) * CPB, W1));
                    fetch_batch(bn);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef CAMD_DBG_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
        }
        return;
    }
    for (int t0 = 0; t0 < nsteps; t0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; u++) {
            const int t = t0 + u;
            {
                {
                    const int xi = t - grp;
                    const bool act = rvalid && xi >= 0 && xi < W1;
                    // ---- prefetch C / S of column xi+RING-1 into the slot released by the previous step
                    // (unconditional: out-of-range columns are clamped and their data ignored)
                    {
                        constexpr int UP = (RING - 1);
                        const int xp = min(max(xi + UP, 0), W1 - 1);
                        uint32_t(&cdst)[NR] = cr[(u + UP) % RING];
                        uint32_t(&sdst)[NR] = sr[(u + UP) % RING];
#ifndef CAMD_DBG_NOLOAD  // measurement variant without the streaming loads
                        load_vec(Crow + cell_off(xp), cdst);
                        if (MODE != 0) load_vec(Srow + cell_off(xp), sdst);
#else
                        (void)xp; (void)cdst; (void)sdst;
#endif
                    }
                    const uint32_t(&cc)[NR] = cr[u];
                    // ---- inputs produced by the row above in the previous step
                    const int rb = (t & 1) ^ 1;
                    uint32_t Vin[NR], Dn[NR], dV = P2pk, dDn = P2pk;
#pragma unroll
                    for (int k = 0; k < NR; k++) { Vin[k] = 0; Dn[k] = 0; }
                    if (grp > 0) {
                        if (HAS_V) {
#pragma unroll
                            for (int v = 0; v < NV; v++) {
                                uint4 w = xV[rb][(threadIdx.x - LANES) * NV + v];
                                Vin[4 * v] = w.x; Vin[4 * v + 1] = w.y; Vin[4 * v + 2] = w.z; Vin[4 * v + 3] = w.w;
                            }
                            dV = xdV[rb][grp - 1];
                        }
                        if (HAS_D) {
#pragma unroll
                            for (int v = 0; v < NV; v++) {
                                uint4 w = xD[rb][(threadIdx.x - LANES) * NV + v];
                                Dn[4 * v] = w.x; Dn[4 * v + 1] = w.y; Dn[4 * v + 2] = w.z; Dn[4 * v + 3] = w.w;
                            }
                            dDn = xdD[rb][grp - 1];
                        }
                    } else if (has_prev && xi < W1) {
                        // record of column xi, parked in the LDS ring by the helper wave
                        const int slot = (xi / CPB) % 3, e = (xi % CPB) * LANES + li;
                        if (HAS_V) {
#pragma unroll
                            for (int v = 0; v < NV; v++) {
                                uint4 w = eVl[slot][e * NV + v];
                                Vin[4 * v] = w.x; Vin[4 * v + 1] = w.y; Vin[4 * v + 2] = w.z; Vin[4 * v + 3] = w.w;
                            }
                            dV = edVl[slot][xi % CPB];
                        }
                        if (HAS_D) {
#pragma unroll
                            for (int v = 0; v < NV; v++) {
                                uint4 w = eDl[slot][e * NV + v];
                                Dn[4 * v] = w.x; Dn[4 * v + 1] = w.y; Dn[4 * v + 2] = w.z; Dn[4 * v + 3] = w.w;
                            }
                            dDn = edDl[slot][xi % CPB];
                        }
                    }

                    uint32_t LVo[NR], LDo[NR], dVo = P2pk, dDo = P2pk;
#pragma unroll
                    for (int k = 0; k < NR; k++) { LVo[k] = 0; LDo[k] = 0; }
                    if (act) {
                        uint32_t s[NR];
#pragma unroll
                        for (int k = 0; k < NR; k++) s[k] = MODE != 0 ? sr[u][k] : 0u;
                        if (HAS_H) {
                            uint32_t L[NR];
                            dH = sgm_step<LANES, NR, PAD>(LH, dH, cc, L, keep, sent, P1pk, P2pk, li);
#pragma unroll
                            for (int k = 0; k < NR; k++) { LH[k] = L[k]; s[k] = pk_addsat_i16(s[k], L[k]); }
                        }
                        if (HAS_V) {
                            dVo = sgm_step<LANES, NR, PAD>(Vin, dV, cc, LVo, keep, sent, P1pk, P2pk, li);
#pragma unroll
                            for (int k = 0; k < NR; k++) s[k] = pk_addsat_i16(s[k], LVo[k]);
                        }
                        if (HAS_D) {
                            uint32_t Din[NR];
                            const bool z = xi == 0;  // previous column is outside the array: zero border state
#pragma unroll
                            for (int k = 0; k < NR; k++) Din[k] = z ? 0u : Dh[k];
                            dDo = sgm_step<LANES, NR, PAD>(Din, z ? P2pk : dDh, cc, LDo, keep, sent, P1pk, P2pk, li);
#pragma unroll
                            for (int k = 0; k < NR; k++) s[k] = pk_addsat_i16(s[k], LDo[k]);
                        }
                        const size_t co = cell_off(xi);
#ifdef CAMD_DBG_NOSTORE  // measurement variant: keep the value live, skip the S store
                        asm volatile("" ::"v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]));
                        if (false) {
#else
                        if (MODE != 2 || a.write_S) {
#endif
                            uint4* sp = reinterpret_cast<uint4*>(Srow + co);
#pragma unroll
                            for (int v = 0; v < NV; v++)
                                sp[v] = make_uint4(s[4 * v], s[4 * v + 1], s[4 * v + 2], s[4 * v + 3]);
                        }
                        if (MODE == 2) {
                            // ---- winner-take-all on the final S of this pixel (bit-exact with k_wta) ----
                            // (1) minS and the smallest d attaining it: min over keys (S << 16 | d); padded
                            //     d >= D hold S = 0x7FFF and never win while a real candidate exists
                            uint32_t key = 0xffffffffu;
#pragma unroll
                            for (int k = 0; k < NR; k++) {
                                key = min(key, (s[k] << 16) | dlo[k]);
                                key = min(key, (s[k] & 0xffff0000u) | (dlo[k] + 1u));
                            }
                            key = group_min_u32_full<LANES>(key);
                            const int minS = (int)(key >> 16), best = (int)(key & 0xffffu);
                            // park S so that lane 0 can pick S[best-1], S[best+1] without a select tree
#pragma unroll
                            for (int v = 0; v < NV; v++)
                                wS[threadIdx.x * NV + v] = make_uint4(s[4 * v], s[4 * v + 1], s[4 * v + 2], s[4 * v + 3]);
                            // (2) uniqueness: S[d]*(100-u) < minS*100 for some |d-best| > 1
                            //     <=>  min over those d of S[d]  <=  T = floor((minS*100 - 1) / (100-u))
                            int T = -1;
                            if (minS > 0) T = (int)__fdiv_rn((float)(minS * 100 - 1), (float)(100 - g.uniq));
                            const int t0 = (int)dlo[0] - (best - 1);  // element offset of this lane from best-1
                            uint32_t far = SENT_PK | 0x80008000u;      // 0xFFFF in both halves
#pragma unroll
                            for (int k = 0; k < NR; k++) {
                                const int tk = t0 + 2 * k;
                                uint32_t ex = ((unsigned)tk < 3u ? 0xffffu : 0u) | ((unsigned)(tk + 1) < 3u ? 0xffff0000u : 0u);
                                far = pk_min_u16(far, s[k] | ex);
                            }
                            const int minfar = (int)(group_min_dup16<LANES>(far) & 0xffffu);
                            if (li == 0 && minS < MAX_COST && minfar > T) {
                                const int x = a.sx > 0 ? xi : W1 - 1 - xi;
                                int d = best;
                                int x2 = x + g.minX1 - d - g.minD;
                                const size_t ro = ((size_t)pair * H + y) * (size_t)g.W;
                                atomicMin(a.keys + ro + x2, ((uint32_t)minS << 16) | (uint32_t)(0xffff - d));
                                if (0 < d && d < g.D - 1) {
                                    const uint16_t* gs = reinterpret_cast<const uint16_t*>(wS) + (size_t)grp * (LANES * 8 * NV);
                                    const int Sm = gs[d - 1], Sp = gs[d + 1];
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
                        }
                    }
                    // ---- hold the Dg input for the next step; publish this step's outputs
#pragma unroll
                    for (int k = 0; k < NR; k++) Dh[k] = Dn[k];
                    dDh = dDn;
                    const int wb = t & 1;
                    if (HAS_V) {
#pragma unroll
                        for (int v = 0; v < NV; v++)
                            xV[wb][threadIdx.x * NV + v] =
                                make_uint4(LVo[4 * v], LVo[4 * v + 1], LVo[4 * v + 2], LVo[4 * v + 3]);
                        if (li == 0) xdV[wb][grp] = dVo;
                    }
                    if (HAS_D) {
#pragma unroll
                        for (int v = 0; v < NV; v++)
                            xD[wb][threadIdx.x * NV + v] =
                                make_uint4(LDo[4 * v], LDo[4 * v + 1], LDo[4 * v + 2], LDo[4 * v + 3]);
                        if (li == 0) xdD[wb][grp] = dDo;
                    }
                    if (producer && act) {
                        unsigned long long* p = Eout + ((size_t)xi * LANES + li) * EREC;
#pragma unroll
                        for (int k = 0; k < 2 * NV; k++)
                            __hip_atomic_store(p + k,
                                               (unsigned long long)LVo[2 * k] | ((unsigned long long)LVo[2 * k + 1] << 32),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int k = 0; k < 2 * NV; k++)
                            __hip_atomic_store(p + 2 * NV + k,
                                               (unsigned long long)LDo[2 * k] | ((unsigned long long)LDo[2 * k + 1] << 32),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p + 4 * NV, (unsigned long long)dVo | ((unsigned long long)dDo << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((xi % BAND_CHUNK) == BAND_CHUNK - 1 || xi == W1 - 1) {
                            // the write-through stores of the whole chunk must have landed before the flag is raised
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (li == 0)
                                __hip_atomic_store(Fout + xi / BAND_CHUNK, a.epoch, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                // LDS-only barrier: __syncthreads() would also drain every outstanding global load/store
                // (vmcnt(0)) and with it the whole C/S prefetch ring
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef CAMD_DBG_NOBARRIER
                __builtin_amdgcn_s_barrier();
#endif
            }
        }
    }
}

// left-right check of one row from the candidates / right-view keys the FINAL band pass produced
__global__ __launch_bounds__(256) void k_lrcheck(const int16_t* __restrict__ d1, const uint32_t* __restrict__ keys,
                                                 int16_t* __restrict__ out, size_t out_pitch_e, size_t out_stride_e,
                                                 Geom g)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const int y = blockIdx.y, pair = blockIdx.z;
    const size_t ro = ((size_t)pair * g.H + y) * (size_t)g.W;
    const int INVALID_SCALED = (g.minD - 1) * 16;
    int v1 = d1[ro + x];
    if (x >= g.minX1 && x < g.minX1 + g.W1 && v1 != INVALID_SCALED) {
        int _d = v1 >> 4, d_ = (v1 + 15) >> 4;
        int _x = x - _d, x_ = x - d_;
        bool bad = false;
        if (0 <= _x && _x < g.W) {
            uint32_t k = keys[ro + _x];
            int v = k == KEY_INIT ? INVALID_SCALED : (int)(0xffffu - (k & 0xffffu)) + g.minD;
            bad = v >= g.minD && abs(v - _d) > g.d12;
        }
        if (bad) {
            bad = false;
            if (0 <= x_ && x_ < g.W) {
                uint32_t k = keys[ro + x_];
                int v = k == KEY_INIT ? INVALID_SCALED : (int)(0xffffu - (k & 0xffffu)) + g.minD;
                bad = v >= g.minD && abs(v - d_) > g.d12;
            }
        }
        if (bad) v1 = INVALID_SCALED;
    }
    out[(size_t)pair * out_stride_e + (size_t)y * out_pitch_e + x] = (int16_t)v1;
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

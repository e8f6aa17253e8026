// common.hpp -- shared helpers for the gfx950 kernels of libcalibrating_amd.so
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/calibrating_amd.h"
#include "../../include/calibrating_amd_experimental.h"  // (the library implements these too; the product never calls them)

namespace camd {

void set_error(const char* fmt, ...);

#define CAMD_HIP(call)                                                                  \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            camd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                            __FILE__, __LINE__);                                        \
            return CAMD_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)

#define CAMD_LAUNCH_CHECK()                                                             \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            camd::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),  \
                            __FILE__, __LINE__);                                        \
            return CAMD_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)

// ---- packed 16-bit arithmetic on a dword (two u16 lanes: lo = even element, hi = odd element) ----
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a),
                                                                  __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a),
                                                                  __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, a) + __builtin_bit_cast(u16x2_t, b));
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, a) - __builtin_bit_cast(u16x2_t, b));
}
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2_t, a),
                                                                  __builtin_bit_cast(s16x2_t, b)));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a),
                                                                  __builtin_bit_cast(s16x2_t, b)));
}
// unsigned saturating subtract: max(a - b, 0) per half
__device__ __forceinline__ uint32_t pk_subsat_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2_t, a),
                                                                      __builtin_bit_cast(u16x2_t, b)));
}
// signed saturating add (saturate_cast<short>)
__device__ __forceinline__ uint32_t pk_addsat_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2_t, a),
                                                                      __builtin_bit_cast(s16x2_t, b)));
}
// signed saturating subtract
__device__ __forceinline__ uint32_t pk_subsat_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_t, a),
                                                                      __builtin_bit_cast(s16x2_t, b)));
}
// a * b + c per half, saturated to 0xFFFF (v_pk_mad_u16 ... clamp)
__device__ __forceinline__ uint32_t pk_mad_sat_u16(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_lshr_u16(uint32_t a, uint32_t sh_pk)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, a) >> __builtin_bit_cast(u16x2_t, sh_pk));
}
// (hi:lo) >> 16 taken from the 64-bit concatenation {hi, lo}: result.lo = lo.hi, result.hi = hi.lo
__device__ __forceinline__ uint32_t alignbit16(uint32_t hi, uint32_t lo)
{
    return __builtin_amdgcn_alignbit(hi, lo, 16);
}
__device__ __forceinline__ uint32_t dup16(uint32_t v) { return (v & 0xffffu) | (v << 16); }

// ---- DPP cross-lane moves (gfx9 encodings). Lanes whose source is outside the row / disabled
//      keep `old` (bound_ctrl = 0). A DPP "row" is 16 lanes. ----
enum : int {
    DPP_QUAD_XOR1 = 0xB1,  // quad_perm:[1,0,3,2]
    DPP_QUAD_XOR2 = 0x4E,  // quad_perm:[2,3,0,1]
    DPP_ROW_SHL1 = 0x101,  // lane i <- lane i+1 (within the row)
    DPP_ROW_SHR1 = 0x111,  // lane i <- lane i-1 (within the row)
    DPP_WAVE_SHL1 = 0x130, // lane i <- lane i+1 (whole wave)
    DPP_WAVE_SHR1 = 0x138, // lane i <- lane i-1 (whole wave)
    DPP_ROW_MIRROR = 0x140,
    DPP_ROW_HALF_MIRROR = 0x141
};

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, 0xf, false);
}

// min over the LANES-lane group (LANES in {1,2,4,8,16}, groups aligned inside a 16-lane row);
// every lane of the group ends up with the group's packed minimum.
template <int LANES>
__device__ __forceinline__ uint32_t group_min_pk_u16(uint32_t v)
{
    if (LANES >= 2) v = pk_min_u16(v, dpp_mov<DPP_QUAD_XOR1>(v, v));
    if (LANES >= 4) v = pk_min_u16(v, dpp_mov<DPP_QUAD_XOR2>(v, v));
    if (LANES >= 8) v = pk_min_u16(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v, v));
    if (LANES >= 16) v = pk_min_u16(v, dpp_mov<DPP_ROW_MIRROR>(v, v));
    return v;
}
template <int LANES>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v)
{
    if (LANES >= 2) v = min(v, dpp_mov<DPP_QUAD_XOR1>(v, v));
    if (LANES >= 4) v = min(v, dpp_mov<DPP_QUAD_XOR2>(v, v));
    if (LANES >= 8) v = min(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v, v));
    if (LANES >= 16) v = min(v, dpp_mov<DPP_ROW_MIRROR>(v, v));
    return v;
}
template <int LANES>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v)
{
    if (LANES >= 2) v = max(v, dpp_mov<DPP_QUAD_XOR1>(v, v));
    if (LANES >= 4) v = max(v, dpp_mov<DPP_QUAD_XOR2>(v, v));
    if (LANES >= 8) v = max(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v, v));
    if (LANES >= 16) v = max(v, dpp_mov<DPP_ROW_MIRROR>(v, v));
    return v;
}
template <int LANES>
__device__ __forceinline__ uint32_t group_or_u32(uint32_t v)
{
    if (LANES >= 2) v |= dpp_mov<DPP_QUAD_XOR1>(v, v);
    if (LANES >= 4) v |= dpp_mov<DPP_QUAD_XOR2>(v, v);
    if (LANES >= 8) v |= dpp_mov<DPP_ROW_HALF_MIRROR>(v, v);
    if (LANES >= 16) v |= dpp_mov<DPP_ROW_MIRROR>(v, v);
    return v;
}

// Butterflies for groups whose lanes are all active: the DPP source lane is always valid, so no `old`
// value has to be preserved and the move folds into the VALU op (v_min_u32_dpp: one instruction per step).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_perm(uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, 0xf, 0xf, true);
}
template <int LANES>
__device__ __forceinline__ uint32_t group_min_u32_full(uint32_t v)
{
    if (LANES >= 2) v = min(v, dpp_perm<DPP_QUAD_XOR1>(v));
    if (LANES >= 4) v = min(v, dpp_perm<DPP_QUAD_XOR2>(v));
    if (LANES >= 8) v = min(v, dpp_perm<DPP_ROW_HALF_MIRROR>(v));
    if (LANES >= 16) v = min(v, dpp_perm<DPP_ROW_MIRROR>(v));
    return v;
}
// minimum over all u16 of the group's packed values, returned in BOTH halves (ready for packed use)
template <int LANES>
__device__ __forceinline__ uint32_t group_min_dup16(uint32_t pk)
{
    uint32_t v = pk_min_u16(pk, alignbit16(pk, pk));  // both halves = min(lo, hi)
    return group_min_u32_full<LANES>(v);              // x * 0x10001 is monotonic in x
}

// The int16 regime the SGBM kernels implement: camd_sgbm_create refuses parameters beyond these (normalise() in
// sgbm.hip), and the kernels' carry-free 32-bit arithmetic on packed 16-bit pairs is proved against the same numbers
// (sgm_step's + P1 / + P2, NOCARRY in sgbm_cost.hpp) -- one definition, so a raised limit cannot leave a stale proof.
constexpr int CAMD_MAX_P2 = 24000;
constexpr int CAMD_MAX_FTZERO = 127;
static_assert(CAMD_MAX_P2 < 0x8000, "P1 < P2 must fit a signed 16-bit half: sgm_step adds them with plain 32-bit adds");
static_assert(2 * CAMD_MAX_FTZERO + 63 < 0x8000, "a pixel cost must fit a 16-bit half");

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace camd

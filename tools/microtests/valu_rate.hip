// Issue rate of the VALU instructions the kernels lean on (gfx950): 8 independent chains per lane, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define OPS(X) \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1") \
    X(1, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") \
    X(2, "v_dot2_i32_i16", "v_dot2_i32_i16 %0, %1, %2, %0") \
    X(3, "v_dot2c_i32_i16", "v_dot2c_i32_i16 %0, %1, %2") \
    X(4, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, %2") \
    X(5, "v_dot4_i32_i8", "v_dot4_i32_i8 %0, %1, %2, %0") \
    X(6, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %1, %2, %0") \
    X(7, "v_dot2_u32_u16", "v_dot2_u32_u16 %0, %1, %2, %0") \
    X(8, "v_pk_sub_u16 clamp", "v_pk_sub_u16 %0, %0, %1 clamp") \
    X(9, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1") \
    X(10, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1") \
    X(11, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %0, %1, %0") \
    X(12, "v_or_b32", "v_or_b32 %0, %0, %1") \
    X(13, "v_mad_u32_u24", "v_mad_u32_u24 %0, %1, %2, %0") \
    X(14, "v_mad_i32_i16", "v_mad_i32_i16 %0, %1, %2, %0") \
    X(15, "v_pk_mad_u16", "v_pk_mad_u16 %0, %1, %2, %0") \
    X(16, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(17, "v_add_u32 dpp wave_shr:1", "v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf") \
    X(18, "v_min_u32", "v_min_u32 %0, %0, %1") \
    X(19, "v_min3_u32", "v_min3_u32 %0, %0, %1, %2") \
    X(20, "v_lshl_or_b32", "v_lshl_or_b32 %0, %1, 16, %0") \
    X(21, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2") \
    X(22, "v_bfe_u32", "v_bfe_u32 %0, %0, 8, 8") \
    X(23, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %1") \
    X(24, "v_fmac_f32", "v_fmac_f32 %0, %1, %2") \
    X(25, "v_pk_fma_f32 (64-bit)", "") \
    X(26, "v_sad_u8", "v_sad_u8 %0, %1, %2, %0") \
    X(27, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2") \
    X(28, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %1") \
    X(29, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
    X(30, "v_bitop3_b32 (a|b)&c", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8") \
    X(31, "v_pk_add_i16 clamp", "v_pk_add_i16 %0, %0, %1 clamp") \
    X(32, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
    X(33, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
    X(34, "v_and_b32", "v_and_b32 %0, %0, %1") \
    X(35, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0") \
    X(36, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0") \
    X(37, "v_max_u32", "v_max_u32 %0, %0, %1") \
    X(38, "v_max_i32", "v_max_i32 %0, %0, %1") \
    X(39, "v_max_f32", "v_max_f32 %0, %0, %1") \
    X(40, "v_min_f32", "v_min_f32 %0, %0, %1") \
    X(41, "v_add_f32", "v_add_f32 %0, %0, %1") \
    X(42, "v_mul_f32", "v_mul_f32 %0, %0, %1") \
    X(43, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(44, "v_mov_b32", "v_mov_b32 %0, %1") \
    X(45, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2") \
    X(46, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1") \
    X(47, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2") \
    X(48, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2") \
    X(49, "v_min_u16", "v_min_u16 %0, %0, %1") \
    X(50, "v_add_u16", "v_add_u16 %0, %0, %1") \
    X(51, "v_sub_u16", "v_sub_u16 %0, %0, %1") \
    X(52, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2") \
    X(53, "v_fma_f32", "v_fma_f32 %0, %1, %2, %0") \
    X(54, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 16") \
    X(55, "v_subrev_u32", "v_subrev_u32 %0, %0, %1") \
    X(56, "v_add_u32 sdwa", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
    X(57, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %0") \
    X(58, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1") \
    X(59, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1") \
    X(60, "v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %1") \
    X(61, "v_not_b32", "v_not_b32 %0, %1") \
    X(62, "v_bfi_b32", "v_bfi_b32 %0, %1, %2, %0") \
    X(63, "v_and_b32 dpp row_shr:1", "v_and_b32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t x[8];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = seed * (j + 1) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
#define X(N, NAME, ASM) if (OP == N && N != 25) asm volatile(ASM : "+v"(x[j]) : "v"(a), "v"(b));
            OPS(X)
#undef X
        }
        if (OP == 25) {
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                f2 v = {__builtin_bit_cast(float, x[j]), __builtin_bit_cast(float, x[j + 1])};
                f2 m = {__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(v) : "v"(m));
                x[j] = __builtin_bit_cast(uint32_t, v.x); x[j + 1] = __builtin_bit_cast(uint32_t, v.y);
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    uint32_t* out; (void)hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
#define X(N, NAME, ASM) { float ms = 0; for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); k_rate<N><<<2048, 256>>>(out, iters, 12345); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); } \
        double insts = 2048.0 * 4 * iters * (N == 25 ? 4 : 8); \
        printf("%-28s %.3f ms  %.2f wave-instr / cycle / CU at 2.4 GHz (full rate = 1.00: one per SIMD every 4 cycles)\n", NAME, ms, insts / (ms * 1e-3) / 2.4e9 / 256); }
    OPS(X)
#undef X
    return 0;
}

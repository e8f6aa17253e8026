// Semantics and issue rate of the byte-SAD family on gfx950, and unaligned LDS reads.
//   hipcc --offload-arch=gfx950 -O3 qsad_rate.hip -o qsad_rate && ./qsad_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>

__global__ void k_sem(const uint64_t* a, const uint32_t* b, const uint64_t* c, uint64_t* o_mqsad, uint64_t* o_qsad,
                      uint32_t* o_msad, uint32_t* o_sad)
{
    int i = threadIdx.x;
    o_mqsad[i] = __builtin_amdgcn_mqsad_pk_u16_u8(a[i], b[i], c[i]);
    o_qsad[i] = __builtin_amdgcn_qsad_pk_u16_u8(a[i], b[i], c[i]);
    o_msad[i] = __builtin_amdgcn_msad_u8((uint32_t)a[i], b[i], (uint32_t)c[i]);
    o_sad[i] = __builtin_amdgcn_sad_u8((uint32_t)a[i], b[i], (uint32_t)c[i]);
}

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint64_t* out, int iters, uint64_t seed)
{
    uint64_t x[8];
    uint32_t r = (uint32_t)seed + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = seed * (j + 1) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) x[j] = __builtin_amdgcn_mqsad_pk_u16_u8(x[j], r, x[(j + 1) & 7]);
            if (OP == 1) x[j] = __builtin_amdgcn_qsad_pk_u16_u8(x[j], r, x[(j + 1) & 7]);
            if (OP == 2) { uint32_t lo = (uint32_t)x[j]; lo = __builtin_amdgcn_sad_u8(lo, r, (uint32_t)x[(j + 1) & 7]); x[j] = (x[j] & 0xffffffff00000000ull) | lo; }
            if (OP == 3) {  // two packed-u16 ops (the 64-bit equivalent of one QSAD result)
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                uint32_t lo = (uint32_t)x[j], hi = (uint32_t)(x[j] >> 32);
                u16x2 a = __builtin_bit_cast(u16x2, lo), b = __builtin_bit_cast(u16x2, hi), c = __builtin_bit_cast(u16x2, r);
                a = __builtin_elementwise_sub_sat(a, c); b = __builtin_elementwise_sub_sat(b, c);
                asm volatile("" : "+v"(a), "+v"(b));
                x[j] = __builtin_bit_cast(uint32_t, a) | ((uint64_t)__builtin_bit_cast(uint32_t, b) << 32);
            }
            if (OP == 4) { uint32_t lo = (uint32_t)x[j]; lo = __builtin_amdgcn_msad_u8(lo, r, (uint32_t)x[(j + 1) & 7]); x[j] = (x[j] & 0xffffffff00000000ull) | lo; }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// unaligned LDS reads: lane l reads 12 bytes at byte offset l*stride + off
__global__ void k_lds(const uint8_t* in, uint32_t* out, int stride, int off, int iters, uint64_t* cycles)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = in[i];
    __syncthreads();
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    const uint8_t* p = s + threadIdx.x * stride + off;
    uint64_t t0 = clock64();
    for (int it = 0; it < iters; it++) {
        uint32_t v[3];
        __builtin_memcpy(v, p + (it & 3) * 16, 12);
        acc0 ^= v[0]; acc1 += v[1]; acc2 ^= v[2];
    }
    uint64_t t1 = clock64();
    uint32_t v[3];
    __builtin_memcpy(v, p, 12);
    out[threadIdx.x * 4 + 0] = v[0]; out[threadIdx.x * 4 + 1] = v[1]; out[threadIdx.x * 4 + 2] = v[2];
    out[threadIdx.x * 4 + 3] = acc0 ^ acc1 ^ acc2;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

static uint32_t sadu8(uint32_t a, uint32_t b, bool masked) {
    uint32_t s = 0;
    for (int i = 0; i < 4; i++) { int x = (a >> (8 * i)) & 255, y = (b >> (8 * i)) & 255; if (masked && y == 0) continue; s += abs(x - y); }
    return s;
}

int main()
{
    const int N = 64;
    uint64_t ha[N], hc[N]; uint32_t hb[N];
    srand(7);
    for (int i = 0; i < N; i++) {
        ha[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand();
        hb[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
        if (i % 3 == 0) hb[i] &= 0xff00ffffu;  // a zero reference byte
        if (i % 5 == 0) hb[i] &= 0x000000ffu;  // only byte 0 set
        if (i % 7 == 0) ha[i] &= 0xffffff00ffffff00ull;  // zero bytes in S0
        hc[i] = (i & 1) ? 0xfff0ffe0ff01fffeull : ((uint64_t)rand() << 20);
    }
    uint64_t *a, *c, *om, *oq; uint32_t *b, *oms, *os;
    hipMalloc(&a, N * 8); hipMalloc(&c, N * 8); hipMalloc(&om, N * 8); hipMalloc(&oq, N * 8);
    hipMalloc(&b, N * 4); hipMalloc(&oms, N * 4); hipMalloc(&os, N * 4);
    hipMemcpy(a, ha, N * 8, hipMemcpyHostToDevice); hipMemcpy(b, hb, N * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc, N * 8, hipMemcpyHostToDevice);
    k_sem<<<1, N>>>(a, b, c, om, oq, oms, os);
    uint64_t rm[N], rq[N]; uint32_t rms[N], rs[N];
    hipMemcpy(rm, om, N * 8, hipMemcpyDeviceToHost); hipMemcpy(rq, oq, N * 8, hipMemcpyDeviceToHost);
    hipMemcpy(rms, oms, N * 4, hipMemcpyDeviceToHost); hipMemcpy(rs, os, N * 4, hipMemcpyDeviceToHost);
    // candidate models: mask on S1 (reference) bytes == 0 / mask on S0 bytes == 0; u16 accumulate wraps / saturates
    int bad_q_wrap = 0, bad_q_sat = 0, bad_m_s1_wrap = 0, bad_m_s1_sat = 0, bad_m_s0_wrap = 0, bad_msad_s1 = 0, bad_msad_s0 = 0, bad_sad = 0;
    for (int i = 0; i < N; i++) {
        uint64_t eq_w = 0, eq_s = 0, em1_w = 0, em1_s = 0, em0_w = 0;
        for (int k = 0; k < 4; k++) {
            uint32_t win = (uint32_t)(ha[i] >> (8 * k));
            uint32_t acc = (hc[i] >> (16 * k)) & 0xffff;
            uint32_t q = sadu8(win, hb[i], false) + acc, m1 = sadu8(win, hb[i], true) + acc;
            uint32_t m0 = 0; for (int j = 0; j < 4; j++) { int x = (win >> (8 * j)) & 255, y = (hb[i] >> (8 * j)) & 255; if (x) m0 += abs(x - y); } m0 += acc;
            eq_w |= (uint64_t)(q & 0xffff) << (16 * k); eq_s |= (uint64_t)(q > 0xffff ? 0xffff : q) << (16 * k);
            em1_w |= (uint64_t)(m1 & 0xffff) << (16 * k); em1_s |= (uint64_t)(m1 > 0xffff ? 0xffff : m1) << (16 * k);
            em0_w |= (uint64_t)(m0 & 0xffff) << (16 * k);
        }
        bad_q_wrap += rq[i] != eq_w; bad_q_sat += rq[i] != eq_s;
        bad_m_s1_wrap += rm[i] != em1_w; bad_m_s1_sat += rm[i] != em1_s; bad_m_s0_wrap += rm[i] != em0_w;
        uint32_t ms0 = (uint32_t)hc[i]; for (int j = 0; j < 4; j++) { int x = ((uint32_t)ha[i] >> (8 * j)) & 255, y = (hb[i] >> (8 * j)) & 255; if (x) ms0 += abs(x - y); }
        bad_msad_s1 += rms[i] != sadu8((uint32_t)ha[i], hb[i], true) + (uint32_t)hc[i];
        bad_msad_s0 += rms[i] != ms0;
        bad_sad += rs[i] != sadu8((uint32_t)ha[i], hb[i], false) + (uint32_t)hc[i];
        if (i < 4) printf("a=%016llx b=%08x c=%016llx  mqsad=%016llx qsad=%016llx msad=%08x sad=%08x\n", (unsigned long long)ha[i], hb[i],
                          (unsigned long long)hc[i], (unsigned long long)rm[i], (unsigned long long)rq[i], rms[i], rs[i]);
    }
    printf("mismatches of 64: qsad wrap-model %d, qsad saturate-model %d | mqsad mask-S1 wrap %d, mask-S1 saturate %d, mask-S0 wrap %d | msad mask-S1 %d mask-S0 %d | sad %d\n",
           bad_q_wrap, bad_q_sat, bad_m_s1_wrap, bad_m_s1_sat, bad_m_s0_wrap, bad_msad_s1, bad_msad_s0, bad_sad);

    // ---- rate: 256 CUs x 8 waves/SIMD, 8 independent chains, instruction count known
    uint64_t* out; hipMalloc(&out, 2048 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[5] = {"v_mqsad_pk_u16_u8", "v_qsad_pk_u16_u8", "v_sad_u8", "2 x v_pk_sub_u16 clamp", "v_msad_u8"};
    for (int op = 0; op < 5; op++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) k_rate<0><<<2048, 256>>>(out, iters, 12345);
            if (op == 1) k_rate<1><<<2048, 256>>>(out, iters, 12345);
            if (op == 2) k_rate<2><<<2048, 256>>>(out, iters, 12345);
            if (op == 3) k_rate<3><<<2048, 256>>>(out, iters, 12345);
            if (op == 4) k_rate<4><<<2048, 256>>>(out, iters, 12345);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double insts = 2048.0 * 4 * iters * 8 * (op == 3 ? 2 : 1);  // wave-instructions of the op under test
        printf("%-26s %.3f ms  -> %.2f wave-instr/cycle/CU at 2.4 GHz (4 = one per SIMD per cycle... full rate for wave64 is 1 per 4 cycles per SIMD = 1.0/CU/cycle)\n",
               names[op], ms, insts / (ms * 1e-3) / 2.4e9 / 256);
    }
    // ---- unaligned LDS reads
    uint8_t hin[16384]; for (int i = 0; i < 16384; i++) hin[i] = (uint8_t)(i * 7 + (i >> 8));
    uint8_t* din; uint32_t* dout; uint64_t* dcy; hipMalloc(&din, 16384); hipMalloc(&dout, 64 * 16); hipMalloc(&dcy, 8);
    hipMemcpy(din, hin, 16384, hipMemcpyHostToDevice);
    int strides[6] = {1, 3, 4, 12, 16, 48};
    for (int si = 0; si < 6; si++)
        for (int off = 0; off < 4; off++) {
            k_lds<<<1, 64>>>(din, dout, strides[si], off, 4096, dcy);
            uint32_t ho[256]; uint64_t cy; hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost); hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; l++) { uint32_t e[3]; memcpy(e, hin + l * strides[si] + off, 12); for (int k = 0; k < 3; k++) bad += e[k] != ho[l * 4 + k]; }
            printf("lds 12-byte read, lane stride %2d B, offset %d: %d bad words, %.1f cycles per read\n", strides[si], off, bad, (double)cy / 4096);
        }
    return 0;
}

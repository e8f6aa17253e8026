#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* A, const uint32_t* B, uint32_t* out) {
    int i = threadIdx.x;
    uint32_t a = A[i], b = B[i], r1, r2, r3, r4;
    asm volatile("s_nop 4\n v_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r1) : "v"(a), "v"(b));
    asm volatile("s_nop 4\n v_subrev_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r2) : "v"(a), "v"(b));
    asm volatile("s_nop 4\n v_sub_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r3) : "v"(a), "v"(b));
    asm volatile("s_nop 4\n v_subrev_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r4) : "v"(a), "v"(b));
    out[i] = r1; out[64 + i] = r2; out[128 + i] = r3; out[192 + i] = r4;
}
int main() {
    uint32_t ha[64], hb[64]; for (int i = 0; i < 64; i++) { ha[i] = 1000 + i * 3; hb[i] = 50000 + i * 7; }
    uint32_t *da, *db, *o; (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&o, 1024);
    (void)hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, o);
    uint32_t r[256]; (void)hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"v_sub_u32_dpp D,a,b wave_shr", "v_subrev_u32_dpp D,a,b wave_shr", "v_sub_u32_dpp D,a,b row_shr", "v_subrev_u32_dpp D,a,b row_shr"};
    for (int t = 0; t < 4; t++) {
        int i = 5;  // a lane away from row starts
        int got = (int)r[t * 64 + i];
        int shr_a = (int)ha[i - 1], shr_b = (int)hb[i - 1], A_ = (int)ha[i], B_ = (int)hb[i];
        const char* what = got == shr_a - B_ ? "shr(a) - b" : got == B_ - shr_a ? "b - shr(a)" : got == shr_b - A_ ? "shr(b) - a" : got == A_ - shr_b ? "a - shr(b)" : "??";
        printf("%-36s lane 5 = %d  -> %s ; lane 16 = %d (shr(a)-b would be %d, b-shr(a) %d)\n", names[t], got, what, (int)r[t * 64 + 16], (int)ha[15] - (int)hb[16], (int)hb[16] - (int)ha[15]);
    }
    return 0;
}

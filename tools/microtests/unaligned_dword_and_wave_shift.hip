#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const uint8_t* p, uint32_t* out, uint32_t* out2) {
    int i = threadIdx.x;
    uint32_t v; __builtin_memcpy(&v, p + i * 3, 4);
    out[i] = v;
    // wave_shl / wave_shr check
    out2[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, i, 0x130, 0xf, 0xf, true) | ((uint32_t)__builtin_amdgcn_update_dpp(0, i, 0x138, 0xf, 0xf, true) << 16);
}
int main() {
    uint8_t h[256]; for (int i = 0; i < 256; i++) h[i] = i;
    uint8_t* d; uint32_t *o, *o2; hipMalloc(&d, 256); hipMalloc(&o, 256); hipMalloc(&o2, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, o2);
    uint32_t r[64], r2[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) { uint32_t e; memcpy(&e, h + i * 3, 4); if (e != r[i]) { bad++; if (bad < 5) printf("lane %d got %08x want %08x\n", i, r[i], e); } }
    printf("unaligned dword loads: %d bad of 64\n", bad);
    for (int i = 0; i < 64; i += 7) printf("lane %d shl->%u shr->%u\n", i, r2[i] & 0xffff, r2[i] >> 16);
    printf("lane 15 shl %u lane 16 shr %u lane 63 shl %u lane 0 shr %u\n", r2[15] & 0xffff, r2[16] >> 16, r2[63] & 0xffff, r2[0] >> 16);
    return 0;
}

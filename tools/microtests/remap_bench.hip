// A/B of the Lanczos remap kernel's window rows per fetch (8 / 4 / 2: 4 / 6 / 7 waves per SIMD) and images per workgroup on an MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value remap_bench.hip -o /tmp/rb && /tmp/rb
// (includes the product source; every variant must give the same bytes as 8 rows per fetch with one image per workgroup;
// the variants of rounds 2-3 that were dropped -- LDS-resident entry, misaligned fetches, prefetch, LDS staging -- are in git history)
#include "../../calibrating_amd/csrc/remap.hip"
#include <vector>
#include <cstring>
namespace camd { void set_error(const char*, ...) {} }
extern "C" int camd_device_ok(void) { return 0; }

template <int VAR>
static float run(const uint8_t* src, int W, int H, const float* mx, const float* my, uint8_t* dst, const int16_t* tab, int batch, int zb, int reps)
{
    dim3 grid(div_up(W, 256), H, div_up(batch, zb)), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_remap_f32<8, 3, VAR>), grid, block, 0, 0, src, W, H, (size_t)W * 3, (size_t)W * H * 3, mx, my, dst, W, H,
                           (size_t)W * 3, (size_t)W * H * 3, tab, 0, batch, zb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv)
{
    const int W = 1920, H = 1080, B = argc > 1 ? atoi(argv[1]) : 64;
    std::vector<uint8_t> hs((size_t)B * W * H * 3);
    uint32_t x = 12345; for (auto& v : hs) { x = x * 1664525u + 1013904223u; v = x >> 24; }
    std::vector<float> hx((size_t)W * H), hy((size_t)W * H);
    for (int y = 0; y < H; y++) for (int xx = 0; xx < W; xx++) {  // a rig-like map: small rotation + radial distortion, some pixels outside
        double u = (xx - 960.0) / 1000.0, v = (y - 540.0) / 1000.0, r2 = u * u + v * v, k = 1 + 0.08 * r2 - 0.01 * r2 * r2;
        double uu = u * k * 0.9998 - v * k * 0.02, vv = v * k * 0.9998 + u * k * 0.02;
        hx[(size_t)y * W + xx] = (float)(uu * 1000.0 + 965.3); hy[(size_t)y * W + xx] = (float)(vv * 1000.0 + 538.7);
    }
    uint8_t *src, *dst, *ref; float *mx, *my;
    hipMalloc(&src, hs.size()); hipMalloc(&dst, hs.size()); hipMalloc(&ref, hs.size()); hipMalloc(&mx, hx.size() * 4); hipMalloc(&my, hx.size() * 4);
    hipMemcpy(src, hs.data(), hs.size(), hipMemcpyHostToDevice); hipMemcpy(mx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(my, hy.data(), hx.size() * 4, hipMemcpyHostToDevice);
    const int16_t *tl, *tb; camd::get_tables(&tl, &tb);
    run<8>(src, W, H, mx, my, ref, tl, B, 1, 1);
    std::vector<uint8_t> hr(hs.size()), hd(hs.size());
    hipMemcpy(hr.data(), ref, hs.size(), hipMemcpyDeviceToHost);
    const int zbs[5] = {1, 4, 8, 16, 64};
#define VARIANT(V) for (int zi = 0; zi < 5; zi++) { hipMemset(dst, 0x5a, hs.size()); float ms = run<V>(src, W, H, mx, my, dst, tl, B, zbs[zi], 3); \
        hipMemcpy(hd.data(), dst, hs.size(), hipMemcpyDeviceToHost); \
        printf("%d rows per fetch  images/group %2d: %.3f ms per %d images  %s\n", V, zbs[zi], ms, B, \
               memcmp(hd.data(), hr.data(), hs.size()) ? "MISMATCH" : "same bytes"); }
    VARIANT(8) VARIANT(4) VARIANT(2)
    return 0;
}

// HBM read rate of many concurrent sequential streams against the bytes each stream requests at a time.
// The row-parallel last pass of the band aggregation reads two volumes as ~50 K streams (one per image row and
// volume), each advancing 256 B (one pixel's disparity vector) per step.  Does a coarser grain read faster?
//   hipcc --offload-arch=gfx950 -O3 stream_granularity.hip -o /tmp/sg && /tmp/sg
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// a 16-lane group owns one stream; per step it reads CHUNK bytes (16 lanes x 16 B x CHUNK/256 requests, the requests
// of a step contiguous); 4 groups per wave, 7 waves per workgroup like k_band<.., FULL = false>
template <int CHUNK>
__global__ __launch_bounds__(448) void k_streams(const uint4* __restrict__ base, size_t stream_stride16, int nsteps, uint32_t* out, int nstreams)
{
    const int grp = (blockIdx.x * 448 + threadIdx.x) / 16, li = threadIdx.x & 15;
    if (grp >= nstreams) return;
    const uint4* p = base + (size_t)grp * stream_stride16 + li;
    uint32_t acc = 0;
    constexpr int R = CHUNK / 256;
    uint4 ring[4][R];
#pragma unroll
    for (int u = 0; u < 3; u++)
#pragma unroll
        for (int r = 0; r < R; r++) ring[u][r] = p[(size_t)(u * R + r) * 16];
    for (int t0 = 0; t0 < nsteps; t0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u;
#pragma unroll
            for (int r = 0; r < R; r++) ring[(u + 3) & 3][r] = p[(size_t)((t + 3) * R + r) * 16];
#pragma unroll
            for (int r = 0; r < R; r++) acc ^= ring[u][r].x + ring[u][r].y + ring[u][r].z + ring[u][r].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const size_t total = 24ull << 30;  // bytes read per launch
    uint4* buf; (void)hipMalloc(&buf, total + (64 << 20));
    (void)hipMemset(buf, 1, total + (64 << 20));
    uint32_t* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int nstreams : {14336, 28672, 57344, 114688}) {
        const size_t per_stream = total / nstreams / 4096 * 4096;
#define RUN(CH) { const int nsteps = (int)(per_stream / CH) / 4 * 4 - 4; float best = 1e9f; for (int rep = 0; rep < 3; rep++) { (void)hipEventRecord(e0); \
            k_streams<CH><<<(nstreams * 16 + 447) / 448, 448>>>(buf, per_stream / 16, nsteps, out, nstreams); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } \
            printf("%6d streams, %4d B per stream and step: %.2f TB/s\n", nstreams, CH, (double)nstreams * nsteps * CH / (best * 1e-3) / 1e12); }
        RUN(256) RUN(512) RUN(1024)
    }
    return 0;
}

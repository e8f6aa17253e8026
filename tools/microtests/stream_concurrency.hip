// Do kernels on two HIP streams really run at the same time?  Each kernel occupies a fraction of the CUs and spins
// for a fixed time; two concurrent ones should take as long as one.  Also: a big-grid short-workgroup kernel
// (k_cost-like) against a small-grid long-workgroup one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void spin(long long ticks, int* sink)
{
    long long t0 = wall_clock64();
    int x = 0;
    while (wall_clock64() - t0 < ticks) x++;
    if (x == -1) *sink = x;
}

static double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main()
{
    int* sink;
    hipMalloc(&sink, 4);
    hipStream_t s[4];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    const long long t10ms = (long long)rate * 10;
    auto run = [&](const char* what, int nstreams, int wgs, int threads, long long ticks) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < nstreams; i++) hipLaunchKernelGGL(spin, dim3(wgs), dim3(threads), 0, s[i], ticks, sink);
        hipDeviceSynchronize();
        printf("%-70s %.2f ms\n", what, ms_since(t0));
    };
    run("warm-up", 1, 64, 256, t10ms);
    run("1 stream : 64 workgroups x 10 ms", 1, 64, 256, t10ms);
    run("2 streams: 64 workgroups x 10 ms each (concurrent = 10 ms)", 2, 64, 256, t10ms);
    run("4 streams: 64 workgroups x 10 ms each", 4, 64, 256, t10ms);
    run("1 stream : 512 workgroups of 512 threads x 10 ms (fills 2 per CU)", 1, 512, 512, t10ms);
    run("2 streams: 512 workgroups of 512 threads x 10 ms each (16 waves/CU each: both fit)", 2, 512, 512, t10ms);
    run("1 stream : 8192 workgroups of 512 threads x 0.1 ms", 1, 8192, 512, t10ms / 100);
    // mixed: big grid of short workgroups on stream 0, 512 long workgroups on stream 1
    {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(8192), dim3(512), 0, s[0], t10ms / 100, sink);
        hipLaunchKernelGGL(spin, dim3(512), dim3(512), 0, s[1], t10ms, sink);
        hipDeviceSynchronize();
        printf("%-70s %.2f ms\n", "8192 x 0.1 ms on stream 0 || 512 x 10 ms on stream 1", ms_since(t0));
    }
    return 0;
}

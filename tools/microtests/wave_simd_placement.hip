// Which SIMD does wave i of a workgroup land on, and where does the NEXT workgroup on the same CU start?
// (decides whether the band kernel's helper wave can be placed so that two co-resident workgroups load the four
// SIMDs of a CU evenly).  Workgroups of NT threads with 53 KB of LDS (two per CU, like k_band), all resident at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

__global__ void probe(unsigned* out, int waves_per_wg, volatile int* go)
{
    extern __shared__ unsigned char lds[];
    lds[threadIdx.x] = 0;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);        // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID[3:0]
    hw = (hw & 0x00ffffffu) | (xcc << 24);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * waves_per_wg + (threadIdx.x >> 6)] = hw;
    // stay resident until the host has seen every workgroup start, so that pairs really share a CU
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000000) {}
}

int main()
{
    for (int nt : {448, 512, 320}) {
        const int wpw = nt / 64, nwg = 512;
        unsigned* d;
        hipMalloc(&d, nwg * wpw * 4);
        hipMemset(d, 0xff, nwg * wpw * 4);
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(nt), 53 * 1024, 0, d, wpw, nullptr);
        hipDeviceSynchronize();
        std::vector<unsigned> h(nwg * wpw);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        // key of a CU: everything in HW_ID above the SIMD / wave fields
        std::map<unsigned, std::vector<int>> cu;  // -> workgroups
        auto cu_key = [](unsigned hw) { return ((hw >> 8) & 0xffu) | ((hw >> 24) << 8); };  // CU_ID, SH_ID, SE_ID + XCC id
        for (int w = 0; w < nwg; w++) cu[cu_key(h[w * wpw])].push_back(w);
        printf("== %d threads (%d waves) per workgroup: %zu distinct CU keys\n", nt, wpw, cu.size());
        std::map<std::string, int> patterns;
        int shown = 0;
        for (auto& kv : cu) {
            std::string s;
            int load[4] = {0, 0, 0, 0};
            for (int w : kv.second) {
                s += "[";
                for (int i = 0; i < wpw; i++) { int simd = (h[w * wpw + i] >> 4) & 3; s += char('0' + simd); load[simd]++; }
                s += "]";
            }
            char buf[64];
            snprintf(buf, sizeof buf, " load %d%d%d%d", load[0], load[1], load[2], load[3]);
            s += buf;
            patterns[s]++;
            if (shown++ < 3) printf("   cu %06x wgs %zu: %s\n", kv.first, kv.second.size(), s.c_str());
        }
        std::vector<std::pair<int, std::string>> v;
        for (auto& p : patterns) v.push_back({p.second, p.first});
        std::sort(v.rbegin(), v.rend());
        for (size_t i = 0; i < v.size() && i < 8; i++) printf("   %4d x %s\n", v[i].first, v[i].second.c_str());
        hipFree(d);
    }
    return 0;
}

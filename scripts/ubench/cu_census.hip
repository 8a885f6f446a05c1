// CU census (diagnostic): launches G workgroups of 512 threads with L bytes of dynamic LDS (one per CU when L > 80 KiB), each records
// (XCC id, HW_ID, start clock, end clock) and spins ~40 us.  A workgroup whose start lies a whole spin behind the others did not find a
// free CU at launch: the grid was larger than what is resident at once -- on such a box a persistent kernel with a static tile walk runs
// one tile-time longer.   build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/cu_census.hip -o scripts/ubench/cu_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };
__global__ __launch_bounds__(512, 2) void census(Rec* r, long long spin) {
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    if (threadIdx.x == 0) smem[0] = 1;
    while ((long long)((unsigned long long)wall_clock64() - t0) < spin) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        Rec o; o.xcc = xcc & 15u; o.hwid = hwid; o.t0 = t0; o.t1 = (unsigned long long)wall_clock64(); r[blockIdx.x] = o;
    }
}
int main(int argc, char** argv) {
    const int lds = argc > 1 ? atoi(argv[1]) : 139280;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int clk_khz = 100000; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    const long long spin = (long long)clk_khz * 40 / 1000;                 // 40 us in wall-clock ticks
    Rec* d; CK(hipMalloc(&d, sizeof(Rec) * 1024));
    for (int G : {ncu, ncu - 8, ncu + 8}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(census, dim3(G), dim3(512), lds, 0, d, spin);
            CK(hipDeviceSynchronize());
            std::vector<Rec> h(G); CK(hipMemcpy(h.data(), d, sizeof(Rec) * G, hipMemcpyDeviceToHost));
            unsigned long long tmin = ~0ull; for (auto& x : h) tmin = std::min(tmin, x.t0);
            std::map<unsigned, int> per_xcc, per_xcc_late; int late = 0;
            for (auto& x : h) { per_xcc[x.xcc]++; if ((long long)(x.t0 - tmin) > spin / 2) { ++late; per_xcc_late[x.xcc]++; } }
            printf("lds %d G %d (CUs %d) rep %d: late starters %d | WGs per XCC:", lds, G, ncu, rep, late);
            for (auto& kv : per_xcc) printf(" %u:%d(%d late)", kv.first, kv.second, per_xcc_late[kv.first]);
            printf("\n");
        }
    }
    return 0;
}

// Decode tail lab (diagnostic): one decoder layer's o_proj -> gate/up -> down for one token, timed as (a) the three vl2_gemv kernels,
// (b) the tail kernel's three phases as three launches (its geometry: one 1024-thread workgroup per CU), (c) the engine (one launch, two
// grid barriers).  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -I videollama2_amd/csrc scripts/ubench/tail_lab.hip -o scripts/ubench/tail_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "k_decode_tail.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static void fill(std::vector<uint16_t>& h, float scale, unsigned seed) {
    srand(seed);
    for (auto& x : h) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; float g = scale * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        unsigned u; memcpy(&u, &g, 4); x = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
}
int main() {
    const int D = 4096, QD = 4096, I = 14336, NL = 8;                 // NL layers' worth of distinct weights (cold weights every launch, as in a decode step)
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int G = pr.multiProcessorCount & ~7;
    std::vector<uint16_t> hwo((size_t)D * QD), hwgu((size_t)2 * I * D), hwd((size_t)D * I), hx(I);
    fill(hwo, 1.f / 64, 1); fill(hwgu, 1.f / 64, 2); fill(hwd, 1.f / 120, 3); fill(hx, 1.f, 4);
    uint16_t *dwo[NL], *dwgu[NL], *dwd[NL], *o, *x0, *x1, *act, *xo, *x1b, *actb, *xob; float* ones; unsigned* bar;
    for (int l = 0; l < NL; ++l) { CK(hipMalloc(&dwo[l], hwo.size() * 2)); CK(hipMalloc(&dwgu[l], hwgu.size() * 2)); CK(hipMalloc(&dwd[l], hwd.size() * 2));
        CK(hipMemcpy(dwo[l], hwo.data(), hwo.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwgu[l], hwgu.data(), hwgu.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dwd[l], hwd.data(), hwd.size() * 2, hipMemcpyHostToDevice)); }
    for (uint16_t** p : {&o, &x0, &x1, &act, &xo, &x1b, &actb, &xob}) { CK(hipMalloc(p, I * 2)); CK(hipMemcpy(*p, hx.data(), I * 2, hipMemcpyHostToDevice)); }
    std::vector<float> h1(D, 1.f); CK(hipMalloc(&ones, D * 4)); CK(hipMemcpy(ones, h1.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&bar, 128)); CK(hipMemset(bar, 0, 128));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto gemv3 = [&](int l) {
        GemvArgs a{dwo[l], o, nullptr, x0, x1, D, QD, QD, 1e-5f, nullptr, 0, 0, 0};
        hipLaunchKernelGGL((gemv_bf16_kernel<false, false, 1>), dim3(D / 4), dim3(256), (size_t)QD * 2, s, a);
        GemvArgs b{dwgu[l], x1, ones, nullptr, act, 2 * I, D, D, 1e-5f, nullptr, 0, 0, 0};
        hipLaunchKernelGGL((gemv_bf16_kernel<true, false, 1>), dim3(I / 4), dim3(256), (size_t)D * 2, s, b);
        GemvArgs c{dwd[l], act, nullptr, x1, xo, D, I, I, 1e-5f, nullptr, 0, 0, 0};
        hipLaunchKernelGGL((gemv_bf16_kernel<false, false, 1>), dim3(D / 4), dim3(256), (size_t)I * 2, s, c);
    };
    auto targs = [&](int l) { return TailArgs{dwo[l], dwgu[l], dwd[l], QD, D, I, o, x0, x1b, actb, xob, D, QD, I, 1e-5f, bar}; };
    auto phases3 = [&](int l) { TailArgs t = targs(l);
        hipLaunchKernelGGL((decode_tail_kernel<1>), dim3(G), dim3(1024), (size_t)I * 2, s, t);
        hipLaunchKernelGGL((decode_tail_kernel<2>), dim3(G), dim3(1024), (size_t)I * 2, s, t);
        hipLaunchKernelGGL((decode_tail_kernel<4>), dim3(G), dim3(1024), (size_t)I * 2, s, t); };
    auto one = [&](int l, int which) { TailArgs t = targs(l);
        if (which == 1) hipLaunchKernelGGL((decode_tail_kernel<1>), dim3(G), dim3(1024), (size_t)I * 2, s, t);
        if (which == 2) hipLaunchKernelGGL((decode_tail_kernel<2>), dim3(G), dim3(1024), (size_t)I * 2, s, t);
        if (which == 4) hipLaunchKernelGGL((decode_tail_kernel<4>), dim3(G), dim3(1024), (size_t)I * 2, s, t); };
    auto engine = [&](int l) { TailArgs t = targs(l); hipLaunchKernelGGL((decode_tail_kernel<7>), dim3(G), dim3(1024), (size_t)I * 2, s, t); };
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 16; ++i) fn(i % NL);
        double best = 1e30;
        for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0, s)); for (int i = 0; i < 64; ++i) fn(i % NL); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms * 1e3 / 64); }
        printf("%-40s %7.1f us per layer\n", name, best); };
    timeit("three vl2_gemv launches", gemv3);
    timeit("tail kernel, phases as 3 launches", phases3);
    timeit("  phase 0 alone (o_proj)", [&](int l) { one(l, 1); });
    timeit("  phase 1 alone (gate/up)", [&](int l) { one(l, 2); });
    timeit("  phase 2 alone (down)", [&](int l) { one(l, 4); });
    timeit("tail engine (one launch, 2 barriers)", engine);
    std::vector<uint16_t> a(D), b(D); CK(hipMemcpy(a.data(), xo, D * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), xob, D * 2, hipMemcpyDeviceToHost));
    unsigned hb[32]; CK(hipMemcpy(hb, bar, 128, hipMemcpyDeviceToHost));
    printf("engine output == three launches: %s; barrier words re-armed: %s; timeout flag %u\n", memcmp(a.data(), b.data(), D * 2) == 0 ? "yes" : "NO",
           (hb[0] | hb[8] | hb[9] | hb[16]) == 0 ? "yes" : "NO", hb[24]);
    return 0;
}

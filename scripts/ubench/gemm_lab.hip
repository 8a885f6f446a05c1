// GEMM kernel lab (diagnostic, not part of the library): A/B of the library's GEMM kernels straight from C++ -- no Python, no
// torch -- so a kernel edit is one hipcc of this file and a few seconds on the GPU box.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -I videollama2_amd/csrc -I scripts/ubench scripts/ubench/gemm_lab.hip -o scripts/ubench/gemm_lab
//   run:   scripts/ubench/gemm_lab [rounds]          (prints one line per shape and variant: us, TFLOP/s, bitwise == gemm4)
// Operands: N(0,1) bf16 activations, N(0,1)/sqrt(K) weights (random data: the chip is power-limited, zero-filled operands
// flatter every kernel by 15-20 %).  Variants are run in interleaved rounds inside ONE process; min over rounds is reported.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include "k_gemm6.h"
#include "k_gemm5.h"            // scripts/ubench/: the one-wave-per-SIMD lab kernel (measured slower, never dispatched by the library)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename K> static void lds_attr(K k, int bytes) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); }

struct Shape { const char* name; int M, N, K; bool swiglu, bias, res; int act; int norm = 0; };   // norm: 1 RMSNorm-, 2 LayerNorm-carrying (row table + column sums)

static void fill_bf16(std::vector<uint16_t>& h, float scale, unsigned seed) {
    srand(seed);
    for (size_t i = 0; i < h.size(); ++i) {
        float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
        float g = scale * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        unsigned u; memcpy(&u, &g, 4); h[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
}

template <int ACT, bool SW>
static void launch(int variant, GemmArgs a, hipStream_t s) {
    a.tile_ctr = nullptr;
    if (variant == 8) {
        lds_attr(gemm4_bf16_kernel<ACT, SW, false>, GEMM4_LDS_BYTES);
        a.tiles_m = (a.M + 255) / 256; a.tiles_n = a.N / 256;
        hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (variant == 9) {                          // gemm4 with the register-resident (C^T) epilogue
        lds_attr(gemm4_bf16_kernel<ACT, SW, false, true>, GEMM4_LDS_BYTES);
        a.tiles_m = (a.M + 255) / 256; a.tiles_n = a.N / 256;
        hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false, true>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (variant == 16) {                         // gemm4 on 192 x 256 tiles, LDS epilogue
        lds_attr(gemm4_bf16_kernel<ACT, SW, false, false, -1, 192>, GEMM4_LDS_BYTES);
        a.tiles_m = (a.M + 191) / 192; a.tiles_n = a.N / 256;
        hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false, false, -1, 192>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (variant == 17) {                         // ... with the register-resident (C^T) epilogue
        lds_attr(gemm4_bf16_kernel<ACT, SW, false, true, -1, 192>, GEMM4_LDS_BYTES);
        a.tiles_m = (a.M + 191) / 192; a.tiles_n = a.N / 256;
        hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false, true, -1, 192>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (variant == 10) {                         // ... with the epilogue flags compiled in (EF)
        a.tiles_m = (a.M + 255) / 256; a.tiles_n = a.N / 256;
        const int ef = (a.bias ? EF_BIAS : 0) | (a.res ? EF_RES : 0);
#define LAB_EF(E) case E: lds_attr(gemm4_bf16_kernel<ACT, SW, false, true, E>, GEMM4_LDS_BYTES); \
        hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false, true, E>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a); break;
        switch (ef) { LAB_EF(0) LAB_EF(1) LAB_EF(8) LAB_EF(9) }
#undef LAB_EF
    } else if (variant == 5) {                          // gemm3 (128 x 256) with the register-resident epilogue, flags compiled in
        a.tiles_m = (a.M + 127) / 128; a.tiles_n = a.N / 256;
        const int ef = (a.bias ? EF_BIAS : 0) | (a.res ? EF_RES : 0);
#define LAB_EF(E) case E: lds_attr(gemm3_bf16_kernel<ACT, SW, false, true, E>, GEMM3_LDS_BYTES); \
        hipLaunchKernelGGL((gemm3_bf16_kernel<ACT, SW, false, true, E>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM3_LDS_BYTES, s, a); break;
        switch (ef) { LAB_EF(0) LAB_EF(1) LAB_EF(8) LAB_EF(9) }
#undef LAB_EF
    } else if (variant == 4) {
        lds_attr(gemm3_bf16_kernel<ACT, SW, false>, GEMM3_LDS_BYTES);
        a.tiles_m = (a.M + 127) / 128; a.tiles_n = a.N / 256;
        hipLaunchKernelGGL((gemm3_bf16_kernel<ACT, SW, false>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM3_LDS_BYTES, s, a);
    } else if (variant == 1) {
        lds_attr(gemm_bf16_kernel<ACT, SW, false, false>, GEMM_LDS_BYTES);
        a.tiles_m = (a.M + 127) / 128; a.tiles_n = a.N / 128;
        hipLaunchKernelGGL((gemm_bf16_kernel<ACT, SW, false, false>), dim3(a.tiles_m * a.tiles_n), dim3(256), GEMM_LDS_BYTES, s, a);
    } else if (variant == 60 || variant == 61 || variant == 62 || variant == 70 || variant == 71 || variant == 80 || variant == 81) {   // gemm6: persistent 256-row / 192-row / 192-row with two accumulator sets; 70 / 71 = 60 / 61 with dynamic tile hand-out
        static unsigned* dctr = nullptr;
        if (!dctr) { CK(hipMalloc(&dctr, 16)); CK(hipMemset(dctr, 0, 16)); }
        a.tile_ctr = variant >= 70 ? dctr : nullptr;
        a.tile_first_dyn = variant >= 80;
        if (variant >= 80) variant -= 20; else if (variant >= 70) variant -= 10;
        static int ncu = 0;
        if (!ncu) { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; if (getenv("LAB_NCU")) ncu = atoi(getenv("LAB_NCU")); }
        const int bm = variant == 60 ? 256 : 192;
        a.tiles_m = (a.M + bm - 1) / bm; a.tiles_n = a.N / 256;
        const int nt = a.tiles_m * a.tiles_n, g = nt < ncu ? nt : ncu;
        if (variant == 60) { lds_attr(gemm6_bf16_kernel<ACT, SW, 256, false>, GEMM6_LDS_BYTES);
            hipLaunchKernelGGL((gemm6_bf16_kernel<ACT, SW, 256, false>), dim3(g), dim3(512), GEMM6_LDS_BYTES, s, a); }
        else if (variant == 61) { lds_attr(gemm6_bf16_kernel<ACT, SW, 192, false>, GEMM6_LDS_BYTES);
            hipLaunchKernelGGL((gemm6_bf16_kernel<ACT, SW, 192, false>), dim3(g), dim3(512), GEMM6_LDS_BYTES, s, a); }
        else { lds_attr(gemm6_bf16_kernel<ACT, SW, 192, true>, GEMM6_LDS_BYTES);
            hipLaunchKernelGGL((gemm6_bf16_kernel<ACT, SW, 192, true>), dim3(g), dim3(512), GEMM6_LDS_BYTES, s, a); }
    } else if (variant == 22) {
        using G = Gemm5Geo<2, 2>;
        lds_attr(gemm5_bf16_kernel<ACT, SW, false, 2, 2>, G::LDS);
        a.tiles_m = (a.M + G::BM - 1) / G::BM; a.tiles_n = a.N / G::BN;
        hipLaunchKernelGGL((gemm5_bf16_kernel<ACT, SW, false, 2, 2>), dim3(a.tiles_m * a.tiles_n), dim3(256), G::LDS, s, a);
    } else if (variant == 14) {
        using G = Gemm5Geo<1, 4>;
        lds_attr(gemm5_bf16_kernel<ACT, SW, false, 1, 4>, G::LDS);
        a.tiles_m = (a.M + G::BM - 1) / G::BM; a.tiles_n = a.N / G::BN;
        hipLaunchKernelGGL((gemm5_bf16_kernel<ACT, SW, false, 1, 4>), dim3(a.tiles_m * a.tiles_n), dim3(256), G::LDS, s, a);
    } else if (variant == 41) {
        using G = Gemm5Geo<4, 1>;
        lds_attr(gemm5_bf16_kernel<ACT, SW, false, 4, 1>, G::LDS);
        a.tiles_m = (a.M + G::BM - 1) / G::BM; a.tiles_n = a.N / G::BN;
        hipLaunchKernelGGL((gemm5_bf16_kernel<ACT, SW, false, 4, 1>), dim3(a.tiles_m * a.tiles_n), dim3(256), G::LDS, s, a);
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 3;
    std::vector<Shape> shapes = {
        {"tiny", 256, 256, 64, false, false, false, 0},
        {"sq_8192x4096x4096", 8192, 4096, 4096, false, false, false, 0},
        {"sq_8192", 8192, 8192, 8192, false, false, false, 0},
        {"sq_4096", 4096, 4096, 4096, false, false, false, 0},
        {"vit_qkv", 9232, 3072, 1024, false, true, false, 0},
        {"vit_wo", 9232, 1024, 1024, false, true, true, 0},
        {"vit_fc1", 9232, 4096, 1024, false, true, false, 1},
        {"vit_fc2", 9232, 1024, 4096, false, true, true, 0},
        {"stc_s1_conv", 9216, 4096, 4096, false, false, false, 0},
        {"stc_s1_b1", 9216, 4096, 1024, false, false, false, 0},
        {"llm_qkv", 1621, 6144, 4096, false, false, false, 0},
        {"llm_wo", 1621, 4096, 4096, false, false, true, 0},
        {"llm_gateup", 1621, 28672, 4096, true, false, false, 0},
        {"llm_down", 1621, 4096, 14336, false, false, true, 0},
        {"vit_qkv_ln", 9232, 3072, 1024, false, true, false, 0, 2},
        {"vit_fc1_ln", 9232, 4096, 1024, false, true, false, 1, 2},
        {"vit_fc1_9216", 9216, 4096, 1024, false, true, false, 1, 2},
        {"llm_qkv_rms", 1621, 6144, 4096, false, false, false, 0, 1},
        {"llm_gateup_rms", 1621, 28672, 4096, true, false, false, 0, 1},
    };
    if (getenv("LAB_SHAPES")) {                       // comma-separated substrings: only shapes whose name contains one of them
        std::vector<Shape> keep; std::string f = getenv("LAB_SHAPES");
        for (const Shape& sh : shapes) { size_t p0 = 0; while (p0 <= f.size()) { size_t p1 = f.find(',', p0); if (p1 == std::string::npos) p1 = f.size();
            if (p1 > p0 && std::string(sh.name).find(f.substr(p0, p1 - p0)) != std::string::npos) { keep.push_back(sh); break; } p0 = p1 + 1; } }
        shapes = keep;
    }
    std::vector<int> variants = {8, 4, 16, 17};
    if (argc > 2) {                                     // e.g. "8,4": only these variants (v8 is the bit reference, keep it first)
        variants.clear();
        for (char* tok = strtok(argv[2], ","); tok; tok = strtok(nullptr, ",")) variants.push_back(atoi(tok));
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, ncol = sh.swiglu ? sh.N / 2 : sh.N, nc = (size_t)sh.M * ncol;
        std::vector<uint16_t> ha(na), hw(nw), hr(nc);
        fill_bf16(ha, 1.f, 1); fill_bf16(hw, 1.f / sqrtf((float)sh.K), 2); fill_bf16(hr, 1.f, 3);
        std::vector<float> hb(sh.N);
        for (int i = 0; i < sh.N; ++i) hb[i] = 0.01f * (float)((i * 37) % 101 - 50);
        std::vector<float> hrn((size_t)sh.M * 2), hcs(sh.N);
        for (int i = 0; i < sh.M; ++i) { hrn[2 * i] = 0.02f * (float)((i * 13) % 41 - 20); hrn[2 * i + 1] = 0.8f + 0.01f * (float)((i * 7) % 37); }
        for (int i = 0; i < sh.N; ++i) hcs[i] = 0.03f * (float)((i * 29) % 53 - 26);
        float *dRN, *dCS;
        CK(hipMalloc(&dRN, hrn.size() * 4)); CK(hipMalloc(&dCS, hcs.size() * 4));
        CK(hipMemcpy(dRN, hrn.data(), hrn.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dCS, hcs.data(), hcs.size() * 4, hipMemcpyHostToDevice));
        uint16_t *dA, *dW, *dR, *dC, *dRef; float* dB;
        CK(hipMalloc(&dA, na * 2)); CK(hipMalloc(&dW, nw * 2)); CK(hipMalloc(&dR, nc * 2)); CK(hipMalloc(&dC, nc * 2)); CK(hipMalloc(&dRef, nc * 2)); CK(hipMalloc(&dB, sh.N * 4));
        CK(hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dR, hr.data(), nc * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
        GemmArgs a{};
        a.A = dA; a.W = dW; a.C = dC; a.bias = sh.bias ? dB : nullptr; a.res = sh.res ? dR : nullptr;
        if (sh.norm) { a.norm = sh.norm; a.norm_eps = 1e-5f; a.row_norm = dRN; if (sh.norm == 2) a.w_colsum = dCS; }
        a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldw = sh.K; a.ldc = (int)ncol; a.ldres = (int)ncol;
        auto go = [&](int v, GemmArgs x) {
            if (sh.swiglu) launch<ACT_NONE, true>(v, x, s);
            else if (sh.act == 1) launch<ACT_QGELU, false>(v, x, s);
            else launch<ACT_NONE, false>(v, x, s);
        };
        std::vector<double> best(variants.size(), 1e30);
        std::vector<int> same(variants.size(), -1);
        std::vector<uint16_t> href(nc), hc(nc);
        for (int r = 0; r < rounds; ++r)
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                const int v = variants[vi];
                if ((v == 14 && sh.N % 512) || ((v == 8 || v == 9 || v == 10 || v == 16 || v == 17 || v >= 60) && sh.N % 256) || (v >= 60 && (sh.res || sh.K < 512)) || ((v == 4 || v == 5) && sh.N % 256) || (v == 22 && sh.N % 256)) continue;
                if (r == 0) {
                    CK(hipMemsetAsync(dC, 0xff, nc * 2, s));
                    go(v, a);
                    CK(hipStreamSynchronize(s));
                    CK(hipGetLastError());
                    CK(hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost));
                    { unsigned long long hsh = 1469598103934665603ull; for (size_t i = 0; i < nc; ++i) { hsh ^= hc[i]; hsh *= 1099511628211ull; }
                      fprintf(stderr, "  hash %s v%d %016llx\n", sh.name, v, hsh); }
                    if (v == 8) href = hc;
                    else { size_t bad = 0; for (size_t i = 0; i < nc; ++i) { if (hc[i] != href[i] && bad < 12 && getenv("LAB_DIFF")) { fprintf(stderr, "    diff at row %zu col %zu: %04x vs %04x; ref holds that value at:", i / ncol, i % ncol, hc[i], href[i]);
                                   int shown = 0; for (size_t q = 0; q < nc && shown < 4; ++q) if (href[q] == hc[i]) { fprintf(stderr, " (%zu,%zu)", q / ncol, q % ncol); ++shown; } fprintf(stderr, "\n"); } bad += hc[i] != href[i]; } same[vi] = bad == 0 ? 1 : 0; if (bad) fprintf(stderr, "  %s v%d: %zu of %zu elements differ from v8\n", sh.name, v, bad, nc); }
                }
                for (int w = 0; w < 2; ++w) go(v, a);
                const int iters = 10;
                if (getenv("LAB_COLD")) {
                    // every timed launch behind a 600 MB memset (Infinity Cache and the L2s hold neither the operands nor the kernel's CODE,
                    // as in the pipeline, where ten other kernels and 100+ MB of other tensors pass between two launches of one GEMM kernel)
                    static void* scratch = nullptr;
                    const size_t sb = (size_t)600 << 20;
                    if (!scratch) CK(hipMalloc(&scratch, sb));
                    double tot = 0;
                    for (int i = 0; i < iters; ++i) {
                        CK(hipMemsetAsync(scratch, i, sb, s));
                        if (getenv("LAB_COLD")[0] == '2') go(8, a);      // '2': another kernel's code between (keeps the operands warm)
                        CK(hipEventRecord(e0, s));
                        go(v, a);
                        CK(hipEventRecord(e1, s));
                        CK(hipEventSynchronize(e1));
                        float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
                        tot += ms1;
                    }
                    best[vi] = std::min(best[vi], tot * 1e3 / iters);
                    continue;
                }
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < iters; ++i) go(v, a);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best[vi] = std::min(best[vi], (double)ms * 1e3 / iters);
            }
        printf("%-18s %5d %5d %5d ", sh.name, sh.M, sh.N, sh.K);
        for (size_t vi = 0; vi < variants.size(); ++vi)
            if (best[vi] < 1e29) printf("| v%-2d %7.1f us %6.0f TF %s ", variants[vi], best[vi], 2.0 * sh.M * sh.N * sh.K / best[vi] / 1e6, same[vi] == 1 ? "==" : (same[vi] == 0 ? "!=" : "  "));
        printf("\n"); fflush(stdout);
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dR)); CK(hipFree(dC)); CK(hipFree(dRef)); CK(hipFree(dB)); CK(hipFree(dRN)); CK(hipFree(dCS));
    }
    return 0;
}

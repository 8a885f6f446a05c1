// Decode-layer lab (diagnostic, round 6): one Mistral-7B decoder layer for ONE token as
//   L0  the product's six launches (qkv GEMV, attention slices, combine, o_proj, gate/up, down),
//   L5  four launches: qkv GEMV -> [RoPE + cache append + attention slices + combine + o_proj in ONE launch, two in-launch hand-offs,
//       the o_proj weights streaming into registers meanwhile] -> gate/up -> down,
// plus the mixed forms and every kernel alone, each captured as a 32-layer hipGraph over distinct weights (cold weights per launch,
// as in a decode step).  Every fused form is compared BIT FOR BIT with the six launches before it is timed.
// build: scripts/ubench/build_lab.sh decode_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include "k_decode.h"
#include "k_decode_fused_lab.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(uint16_t* p, size_t n, float scale, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float u = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.f * scale;
        unsigned b = __builtin_bit_cast(unsigned, u);
        p[i] = (uint16_t)((b + 0x7fff + ((b >> 16) & 1)) >> 16);
    }
}
__global__ void zero_kernel(int* p, int n) { for (int i = threadIdx.x; i < n; i += 256) p[i] = 0; }
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

static uint16_t* dalloc_fill(size_t n, float scale, unsigned seed) {
    uint16_t* p; CK(hipMalloc(&p, n * 2));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, p, n, scale, seed);
    return p;
}

int main(int argc, char** argv) {
    const int D = 4096, NH = 32, NKV = 8, GROUP = 4, QD = NH * 128, KVD = NKV * 128, NQ = QD + 2 * KVD, I = 14336, NL = 32;
    const int SMAX = 2048;
    int POS = argc > 1 ? atoi(argv[1]) : 1650;
    const float eps = 1e-5f, scale_log2e = 0.08838834764831845f * 1.4426950408889634f;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, %d CUs; pos %d (ctx %d, %d slices of 64 keys)\n", pr.name, pr.multiProcessorCount, POS, POS + 1, (POS + 64) / 64);
    std::vector<uint16_t*> wqkv(NL), wo(NL), wgu(NL), wd(NL), kc(NL), vc(NL);
    for (int l = 0; l < NL; ++l) {
        wqkv[l] = dalloc_fill((size_t)NQ * D, 1.f / 64, 11 + l); wo[l] = dalloc_fill((size_t)D * QD, 1.f / 64, 111 + l);
        wgu[l] = dalloc_fill((size_t)2 * I * D, 1.f / 64, 211 + l); wd[l] = dalloc_fill((size_t)D * I, 1.f / 120, 311 + l);
        kc[l] = dalloc_fill((size_t)NKV * SMAX * 128, 1.f, 411 + l); vc[l] = dalloc_fill((size_t)NKV * SMAX * 128, 1.f, 511 + l);
    }
    // two sets of activations / caches: A = six launches, B = the form under test (compared bit for bit)
    uint16_t *x0 = dalloc_fill(D, 1.f, 7), *qkvA, *qkvB, *oA, *oB, *x1A, *x1B, *actA, *actB, *xoA, *xoB;
    for (uint16_t** p : {&qkvA, &qkvB, &oA, &oB, &x1A, &x1B, &actA, &actB, &xoA, &xoB}) { CK(hipMalloc(p, (size_t)I * 2)); CK(hipMemset(*p, 0, (size_t)I * 2)); }
    uint16_t *kcB, *vcB; CK(hipMalloc(&kcB, (size_t)NKV * SMAX * 256)); CK(hipMalloc(&vcB, (size_t)NKV * SMAX * 256));
    const int NSPLIT = (SMAX + 63) / 64;
    float *partA, *partB, *cos_t, *sin_t; CK(hipMalloc(&partA, (size_t)NH * NSPLIT * 130 * 4)); CK(hipMalloc(&partB, (size_t)NH * NSPLIT * 130 * 4));
    CK(hipMemset(partA, 0xff, (size_t)NH * NSPLIT * 130 * 4)); CK(hipMemset(partB, 0xff, (size_t)NH * NSPLIT * 130 * 4));
    { std::vector<float> hc((size_t)SMAX * 64), hs((size_t)SMAX * 64);
      for (int p = 0; p < SMAX; ++p) for (int i = 0; i < 64; ++i) { const double f = p * pow(1e6, -2.0 * i / 128); hc[(size_t)p * 64 + i] = (float)cos(f); hs[(size_t)p * 64 + i] = (float)sin(f); }
      CK(hipMalloc(&cos_t, hc.size() * 4)); CK(hipMalloc(&sin_t, hs.size() * 4));
      CK(hipMemcpy(cos_t, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sin_t, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
    int *pos_dev, *cnt, *err; CK(hipMalloc(&pos_dev, 4)); CK(hipMemcpy(pos_dev, &POS, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&cnt, (NL * (NKV + 1) + 8) * 4)); CK(hipMemset(cnt, 0, (NL * (NKV + 1) + 8) * 4)); err = cnt + NL * (NKV + 1);
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());

    long long* stamps; CK(hipMalloc(&stamps, 4096 * 16 * 8)); CK(hipMemset(stamps, 0, 4096 * 16 * 8)); bool stamps_on = false; int wdelay = argc > 2 ? atoi(argv[2]) : 0;
    // ---- the launches
    auto k_qkv = [&](int l, uint16_t* qkv) {
        GemvArgs a{wqkv[l], x0, nullptr, nullptr, qkv, NQ, D, D, eps, nullptr, 0, 0, 0, 1};
        hipLaunchKernelGGL((gemv_bf16_kernel<false, false, 1>), dim3(NQ / 4), dim3(256), (size_t)D * 2, s, a); };
    auto k_attn = [&](int l, uint16_t* qkv, uint16_t* K, uint16_t* V, float* part) {
        hipLaunchKernelGGL(attn_decode_kernel<false>, dim3(NSPLIT, NKV, 1), dim3(256), 0, s, (const bf16_t*)qkv, K, V, cos_t, sin_t, part, NH, GROUP, NKV, SMAX, 0,
                           (const int*)pos_dev, scale_log2e, 0L, 0L, 0L, (int*)nullptr, (bf16_t*)nullptr); };
    auto k_attnf = [&](int l, uint16_t* qkv, uint16_t* K, uint16_t* V, float* part, uint16_t* o) {
        hipLaunchKernelGGL(attn_decode_kernel<true>, dim3(NSPLIT, NKV, 1), dim3(256), 0, s, (const bf16_t*)qkv, K, V, cos_t, sin_t, part, NH, GROUP, NKV, SMAX, 0,
                           (const int*)pos_dev, scale_log2e, 0L, 0L, 0L, cnt + l * (NKV + 1), o); };
    auto k_comb = [&](float* part, uint16_t* o) {
        hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(NH), dim3(128), 0, s, (const float*)part, o, NSPLIT, 0, (const int*)pos_dev, 0L, 0L); };
    auto k_o = [&](int l, uint16_t* o, uint16_t* x1) {
        GemvArgs a{wo[l], o, nullptr, x0, x1, D, QD, QD, eps, nullptr, 0, 0, 0, 0};
        hipLaunchKernelGGL((gemv_bf16_kernel<false, false, 1>), dim3(D / 4), dim3(256), (size_t)QD * 2, s, a); };
    auto k_gu = [&](int l, uint16_t* x1, uint16_t* act) {
        GemvArgs a{wgu[l], x1, nullptr, nullptr, act, 2 * I, D, D, eps, nullptr, 0, 0, 0, 1};
        hipLaunchKernelGGL((gemv_bf16_kernel<true, false, 1>), dim3(I / 4), dim3(256), (size_t)D * 2, s, a); };
    auto k_down = [&](int l, uint16_t* act, uint16_t* x1, uint16_t* xo) {
        GemvArgs a{wd[l], act, nullptr, x1, xo, D, I, I, eps, nullptr, 0, 0, 0, 0};
        hipLaunchKernelGGL((gemv_bf16_kernel<false, false, 1>), dim3(D / 4), dim3(256), (size_t)I * 2, s, a); };
    // x-first forms of the GEMVs (k_decode2.h gemv_xfirst_bf16_kernel)
    auto k2_qkv = [&](int l, uint16_t* qkv) {
        GemvArgs a{wqkv[l], x0, nullptr, nullptr, qkv, NQ, D, D, eps, nullptr, 0, 0, 0, 1};
        hipLaunchKernelGGL((gemv_xfirst_bf16_kernel<false, false, 2>), dim3(NQ / 4), dim3(256), (size_t)D * 2, s, a); };
    auto k2_o = [&](int l, uint16_t* o, uint16_t* x1) {
        GemvArgs a{wo[l], o, nullptr, x0, x1, D, QD, QD, eps, nullptr, 0, 0, 0, 0};
        hipLaunchKernelGGL((gemv_xfirst_bf16_kernel<false, false, 2>), dim3(D / 4), dim3(256), (size_t)QD * 2, s, a); };
    auto k2_gu = [&](int l, uint16_t* x1, uint16_t* act) {
        GemvArgs a{wgu[l], x1, nullptr, nullptr, act, 2 * I, D, D, eps, nullptr, 0, 0, 0, 1};
        hipLaunchKernelGGL((gemv_xfirst_bf16_kernel<true, false, 2>), dim3(I / 4), dim3(256), (size_t)D * 2, s, a); };
    auto k2_down = [&](int l, uint16_t* act, uint16_t* x1, uint16_t* xo) {
        GemvArgs a{wd[l], act, nullptr, x1, xo, D, I, I, eps, nullptr, 0, 0, 0, 0};
        hipLaunchKernelGGL((gemv_bf16_kernel<false, false, 1>), dim3(D / 4), dim3(256), (size_t)I * 2, s, a); };
    // fused: RoPE + append + attention slices + combine + o_proj (k_decode2.h attn_oproj_kernel)
    auto k_ao = [&](int l, uint16_t* qkv, uint16_t* K, uint16_t* V, float* part, uint16_t* o, uint16_t* x1) {
        AttnOprojArgs a{qkv, K, V, cos_t, sin_t, part, o, wo[l], x0, x1, cnt + l * (NKV + 1), err, pos_dev, NH, GROUP, NKV, SMAX, NSPLIT, D, QD, QD, scale_log2e, wdelay, stamps_on ? stamps : nullptr};
        hipLaunchKernelGGL((attn_oproj_kernel<7>), dim3(D / 16), dim3(512), (size_t)ATTN_OPROJ_DYN_LDS, s, a); };
    auto k_zero = [&]() { hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, s, cnt, NL * (NKV + 1)); };

    // ---- correctness: six launches (A) against every fused piece (B), bit for bit
    auto same = [&](const void* a, const void* b, size_t bytes, const char* what) {
        std::vector<unsigned char> ha(bytes), hb(bytes);
        CK(hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0; for (size_t i = 0; i < bytes; ++i) if (ha[i] != hb[i]) { if (!bad) first = i; ++bad; }
        printf("  %-46s %s", what, bad ? "DIFFERENT" : "equal");
        if (bad) printf(" (%zu bytes, first at %zu)", bad, first);
        printf("\n");
        return bad == 0; };
    bool ok = true;
    {
        const int l = 3;
        CK(hipMemcpy(kcB, kc[l], (size_t)NKV * SMAX * 256, hipMemcpyDeviceToDevice)); CK(hipMemcpy(vcB, vc[l], (size_t)NKV * SMAX * 256, hipMemcpyDeviceToDevice));
        k_qkv(l, qkvA); k_attn(l, qkvA, kc[l], vc[l], partA); k_comb(partA, oA); k_o(l, oA, x1A); k_gu(l, x1A, actA); k_down(l, actA, x1A, xoA);
        k_zero(); k2_qkv(l, qkvB); k_ao(l, qkvB, kcB, vcB, partB, oB, x1B); k_gu(l, x1B, actB); k_down(l, actB, x1B, xoB);
        CK(hipStreamSynchronize(s));
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("bit-identity of the four launches against the six (spin give-up flag %d):\n", herr);
        ok &= same(qkvA, qkvB, (size_t)NQ * 2, "q/k/v rows (x-first GEMV)");
        ok &= same(kc[l], kcB, (size_t)NKV * SMAX * 256, "K cache (appended row)");
        ok &= same(vc[l], vcB, (size_t)NKV * SMAX * 256, "V cache (appended row)");
        const int live = (POS + 64) / 64;
        bool pe = true;
        { std::vector<float> ha((size_t)NH * NSPLIT * 130), hb(ha.size());
          CK(hipMemcpy(ha.data(), partA, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), partB, hb.size() * 4, hipMemcpyDeviceToHost));
          size_t bad = 0; for (int h = 0; h < NH; ++h) for (int i = 0; i < live; ++i) bad += memcmp(&ha[((size_t)h * NSPLIT + i) * 130], &hb[((size_t)h * NSPLIT + i) * 130], 520) != 0;
          pe = bad == 0; printf("  %-46s %s (%zu slices differ)\n", "attention partials of the live slices", pe ? "equal" : "DIFFERENT", bad); }
        ok &= pe;
        ok &= same(oA, oB, (size_t)QD * 2, "attention output o (combine)");
        ok &= same(x1A, x1B, (size_t)D * 2, "x1 = x + o_proj(o)");
        ok &= same(xoA, xoB, (size_t)D * 2, "layer output");
    }
    printf("%s\n", ok ? "ALL EQUAL" : "MISMATCH");
    {   // phase stamps of the fused launch (100 MHz real-time counter), third of three back-to-back launches on distinct layers
        for (int rep = 0; rep < 3; ++rep) { k_zero(); k_qkv(5 + rep, qkvA); stamps_on = rep == 2; k_ao(5 + rep, qkvA, kc[5 + rep], vc[5 + rep], partB, oB, x1B); }
        CK(hipStreamSynchronize(s)); stamps_on = false;
        const int G = D / 16;
        std::vector<long long> st((size_t)G * 16); CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = st[0]; for (int b = 0; b < G; ++b) t0 = std::min(t0, st[(size_t)b * 16]);
        const char* names[11] = {"start", "task: q roped, K/V requested", "task: slice computed, partials stored", "task: drained + barrier", "task: arrival posted",
                                 "combiner: slices arrived", "combiner: o published", "all: o ready (wait over)", "all: o in LDS", "all: rows stored (wave 0)", "all: rows stored (wave 4)"};
        for (int i = 0; i < 11; ++i) {
            double mn = 1e30, mx = 0, sum = 0; int n = 0;
            for (int b = 0; b < G; ++b) { const long long v = st[(size_t)b * 16 + i]; if (!v) continue; const double d = (double)(v - t0) * 0.01; mn = std::min(mn, d); mx = std::max(mx, d); sum += d; ++n; }
            printf("  stamp %2d %-40s n %4d  min %6.2f  mean %6.2f  max %6.2f us\n", i, names[i], n, mn, n ? sum / n : 0, mx);
        }
    }
    // ---- timing: a form = a function enqueuing one layer; captured NL layers deep, replayed
    auto timeit = [&](const char* name, auto fn, int per = 1) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        k_zero();
        for (int l = 0; l < NL; ++l) fn(l);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double best = 1e30, sum = 0;
        const int R = 10;
        for (int r = 0; r < R; ++r) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms); sum += ms; }
        printf("%-64s %7.2f us per %s (best; mean %.2f)\n", name, best * 1e3 / (NL * per), per == 1 ? "layer" : "launch", sum / R * 1e3 / (NL * per));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); };
    auto timeit_eager = [&](const char* name, auto fn) {
        for (int i = 0; i < 2; ++i) { k_zero(); for (int l = 0; l < NL; ++l) fn(l); }
        CK(hipStreamSynchronize(s));
        double best = 1e30, sum = 0;
        const int R = 10;
        for (int r = 0; r < R; ++r) { CK(hipEventRecord(e0, s)); k_zero(); for (int l = 0; l < NL; ++l) fn(l); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms); sum += ms; }
        printf("%-64s %7.2f us per layer (best; mean %.2f)  [EAGER launches]\n", name, best * 1e3 / NL, sum / R * 1e3 / NL); };
    auto six = [&](int l) { k_qkv(l, qkvA); k_attn(l, qkvA, kc[l], vc[l], partA); k_comb(partA, oA); k_o(l, oA, x1A); k_gu(l, x1A, actA); k_down(l, actA, x1A, xoA); };
    for (int rep = 0; rep < 2; ++rep) {
        printf("---- pass %d\n", rep);
        timeit("L0 six launches (product)", six);
        timeit_eager("L0 six launches (product)", six);
        timeit_eager("  empty kernel, 1 WG", [&](int l) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr); });
        timeit_eager("  o_proj GEMV alone", [&](int l) { k_o(l, oA, x1A); });
        timeit("L0q six launches, x-first qkv GEMV", [&](int l) { k2_qkv(l, qkvA); k_attn(l, qkvA, kc[l], vc[l], partA); k_comb(partA, oA); k_o(l, oA, x1A); k_gu(l, x1A, actA); k_down(l, actA, x1A, xoA); });
        timeit("L5 qkv [attn+comb+o] gu down (4), x-first qkv", [&](int l) { k2_qkv(l, qkvB); k_ao(l, qkvB, kc[l], vc[l], partB, oB, x1B); k_gu(l, x1B, actB); k_down(l, actB, x1B, xoB); });
        timeit("  qkv GEMV alone", [&](int l) { k_qkv(l, qkvA); });
        timeit("  qkv GEMV alone, x-first", [&](int l) { k2_qkv(l, qkvA); });
        timeit("  attention slices alone", [&](int l) { k_attn(l, qkvA, kc[l], vc[l], partA); });
        timeit("  attention + elected combine (attn_decode_kernel<true>)", [&](int l) { k_attnf(l, qkvA, kc[l], vc[l], partB, oB); });
        timeit("  combine alone", [&](int l) { k_comb(partA, oA); });
        timeit("  o_proj GEMV alone", [&](int l) { k_o(l, oA, x1A); });
        timeit("  o_proj GEMV alone, x-first", [&](int l) { k2_o(l, oA, x1A); });
        timeit("  gate/up GEMV alone", [&](int l) { k_gu(l, x1A, actA); });
        timeit("  gate/up GEMV alone, x-first", [&](int l) { k2_gu(l, x1A, actA); });
        timeit("  down GEMV alone", [&](int l) { k_down(l, actA, x1A, xoA); });
        timeit("  down GEMV alone, x-first", [&](int l) { k2_down(l, actA, x1A, xoA); });
        timeit("  [attn+comb+o] alone", [&](int l) { k_ao(l, qkvA, kc[l], vc[l], partB, oB, x1B); });
        timeit("  empty kernel, 1 WG", [&](int l) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr); });
        timeit("  empty kernel, 1024 WGs x 256", [&](int l) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, s, (int*)nullptr); });
    }
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("spin give-up flag after the timed runs: %d\n", herr);
    return ok ? 0 : 1;
}

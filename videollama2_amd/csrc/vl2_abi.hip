// libvl2hip.so: extern "C" launchers (include/vl2hip.h) over the gfx950 kernels in k_*.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC vl2_abi.hip -o libvl2hip.so   (see build.py)
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/vl2hip.h"
#include "k_attn.h"
#include "k_decode.h"
#include "k_gemm.h"
#include "k_norm.h"
#include "k_skinny.h"
#include "k_stc.h"
#include "k_vit.h"

static thread_local char g_err[512] = "";
static int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
// set-up calls in front of a launch (LDS opt-in, flag re-arming): the first failure is kept and reported by `launched`
static hipError_t g_setup_err = hipSuccess;
static inline void check(hipError_t e) { if (e != hipSuccess && g_setup_err == hipSuccess) g_setup_err = e; }
static int32_t launched(const char* what) {
    const hipError_t last = hipGetLastError();
    const hipError_t e = g_setup_err != hipSuccess ? g_setup_err : last;
    g_setup_err = hipSuccess;
    if (e != hipSuccess) return fail((int32_t)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}
#define ST(s) ((hipStream_t)(s))
#define ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int32_t vl2_version(void) { return VL2_ABI_VERSION; }
extern "C" const char* vl2_last_error_string(void) { return g_err; }

static int g_gemm_variant = 0;
static void* g_ws = nullptr;          // caller-owned device workspace (vl2_set_workspace)
static int64_t g_ws_bytes = 0;
#define SK_GRID 512                  // persistent stream-K workgroups: 2 per CU
#define SPLITK_MAX_WG 1024           // split-K: at most this many (tile, split) workgroups -> 64 MiB of fp32 partials
#define SPLITK_MAX_TILES 192         // split-K only when the plain grid leaves most of the 512 resident slots empty
// workspace layout: [SPLITK_MAX_WG][64][256] fp32 partial tiles (stream-K uses the first SK_GRID) | stream-K flags
// [SK_GRID + 1] | split-K tile counters [SPLITK_MAX_TILES] (zero when attached, re-armed by the kernel itself)
#define SK_FLAGS_OFF ((int64_t)SPLITK_MAX_WG * 64 * 256 * 4)
#define SPLITK_CNT_OFF (SK_FLAGS_OFF + (int64_t)(SK_GRID + 1) * 4 + 12)
#define SK_WS_BYTES (SPLITK_CNT_OFF + (int64_t)SPLITK_MAX_TILES * 4)
static int g_splitk = 0;             // VL2_TUNE_SPLITK: 0 = never (default: results independent of M), 1 = small grids split K
static int g_attn_kv_groups = 0; // VL2_TUNE_ATTN_KV_GROUPS
static int g_gemv_mr_rpw = 2;   // batched GEMV rows per wave (VL2_TUNE_GEMV_MR_ROWS_PER_WAVE); measured at B=4: 5.56 / 5.04 / 5.56 ms per step at 1 / 2 / 4
static int g_gemv_rpw = 1;   // measured on MI355X: 3.09 / 3.30 / 3.97 ms per 7B decode token at 1 / 2 / 4 rows per wave
extern "C" int32_t vl2_set_tuning(int32_t key, int32_t value) {
    if (key == VL2_TUNE_GEMM_VARIANT && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 32 || value == 256)) { g_gemm_variant = value; return 0; }
    if (key == VL2_TUNE_GEMV_ROWS_PER_WAVE && (value == 1 || value == 2 || value == 4)) { g_gemv_rpw = value; return 0; }
    if (key == VL2_TUNE_SPLITK && (value == 0 || value == 1)) { g_splitk = value; return 0; }
    if (key == VL2_TUNE_GEMV_MR_ROWS_PER_WAVE && (value == 1 || value == 2 || value == 4)) { g_gemv_mr_rpw = value; return 0; }
    if (key == VL2_TUNE_ATTN_KV_GROUPS && value >= 0 && value <= 2) { g_attn_kv_groups = value; return 0; }
    return fail(VL2_E_BADARG, "vl2_set_tuning: unknown key/value %d/%d", key, value);
}

extern "C" int64_t vl2_workspace_bytes(void) { return SK_WS_BYTES; }
extern "C" int32_t vl2_set_workspace(void* ws, int64_t bytes) {
    if (ws && (bytes < SK_WS_BYTES || !ALIGNED16(ws))) return fail(VL2_E_BADARG, "vl2_set_workspace: need >= %lld bytes, 16-byte aligned", (long long)SK_WS_BYTES);
    g_ws = ws;
    g_ws_bytes = ws ? bytes : 0;
    return 0;
}

// ------------------------------------------------------------------------------------------------ GEMM
// stream-K form: measured 0.5-0.6x of the plain grid on this workload's shapes (per-tile prologue/epilogue of the persistent
// workgroups, ~15 us of partial-tile exchange, worse L2 locality of strided tile ownership) -> only on explicit request.
static bool want_stream_k(const GemmArgs&) { return g_ws && g_gemm_variant == 2; }

// Tile-shape choice (auto): expected efficiency = how full the last round of resident workgroups is x rows wasted by the
// M edge x the kernel's measured rate on well-quantised shapes (128x128 two-barrier kernel 1.0, 128x256 ping-pong 1.07,
// 256x256 ping-pong 1.2: profiles/r01_gemm_experiments.md).  Fitted to the measured shapes of the T=16 workload: the LLM
// o/gate-up/down projections and the STC 4096x4096 convs take 128x256, ViT qkv/wo/fc2 and the LLM qkv take 256x256,
// short-K GEMMs (ViT fc1, STC b1) stay on 128x128.  Returns 1, 4 (gemm3) or 8 (gemm4).
static int choose_gemm_kernel(const GemmArgs& a) {
    if (a.N % 256) return 1;
    // at most one 128x128 tile per CU: a bigger tile only halves the CUs in use and doubles the latency of the single round
    // (measured, T=8: 845x4096x4096 44.6 us here vs 50.2 us on 128x256; 945x4096x4096 47.2 vs 51.2)
    if ((long)((a.M + 127) / 128) * (a.N / 128) <= 256) return 1;
    const auto fill = [](double rounds) { return rounds / (double)(long)(rounds + 0.999999); };
    const double m128 = (double)a.M / (((a.M + 127) / 128) * 128.0), m256 = (double)a.M / (((a.M + 255) / 256) * 256.0);
    const double e1 = fill((double)((a.M + 127) / 128) * (a.N / 128) / 512.0) * m128;           // 2 WG/CU
    const double e3 = fill((double)((a.M + 127) / 128) * (a.N / 256) / 256.0) * m128 * 1.07;    // 1 WG/CU
    const double e4 = fill((double)((a.M + 255) / 256) * (a.N / 256) / 256.0) * m256 * 1.2;     // 1 WG/CU
    int best = 1;
    double eb = 1.03 * e1;
    if (a.K >= 2048 && e3 > eb) { best = 4; eb = e3; }
    if (a.K >= 512 && e4 > eb) { best = 8; eb = e4; }
    return best;
}

// Split-K factor for the 128x128 kernel (1 = do not split).  Only for grids that leave most resident slots empty: a lone
// workgroup streams its K-tiles at ~0.64 us each (0.95 us when two share a CU), so a 96-tile grid with K = 4096 takes 42 us
// however idle the chip is (scripts/kernel_bench.py --small).  Cost model in us per launch, d over the divisors of the
// K-tile count: (K-tiles / d) x per-tile time at the resulting occupancy + partial write/reduce.
static int choose_splitk(const GemmArgs& a) {
    if (!g_ws || !g_splitk || (g_gemm_variant != 0 && g_gemm_variant != 1)) return 1;
    const int tiles = a.tiles_m * a.tiles_n, nt = a.K / GEMM_BK;
    if (tiles > SPLITK_MAX_TILES || nt < 32) return 1;       // measured: K = 1024 GEMMs lose (14.0 -> 19.7 us)
    int best = 1;
    double cb = 1e30;
    for (int d = 1; d <= 32 && d * tiles <= SPLITK_MAX_WG; ++d) {
        if (nt % d || nt / d < 4) continue;
        const int wg = tiles * d;
        const double per = wg <= 256 ? 0.64 : 0.95 * ((wg + 511) / 512);
        const double c = (nt / d) * per + (d > 1 ? 2.0 + 0.3 * d : 0.0);
        if (c < cb * (d > 1 ? 0.85 : 1.0)) { cb = c; best = d; }
    }
    return best;
}

// Small-M form (64x64 tiles, gemm_s_bf16_kernel): when the 128x128 grid cannot even give every CU one tile, quartering the
// tile spreads the operand stream over the idle CUs (a workgroup streams at ~55 GB/s whatever the chip does).  Same K order
// as every other kernel -> same bits.  Measured crossover (scripts/kernel_bench.py --small): see profiles/.
static bool want_small_m(const GemmArgs& a) {
    if (g_gemm_variant == 32) return true;
    if (g_gemm_variant != 0) return false;
    return a.tiles_m * a.tiles_n <= 128 && a.K >= 512;   // measured crossover: wins at <= 128 tiles, loses at 152-160
}

template <int ACT, bool SW, bool F32>
static void launch_gemm4(const GemmArgs& a0, hipStream_t s) {
    static bool attr4 = false;
    if (!attr4) {
        check(hipFuncSetAttribute((const void*)gemm4_bf16_kernel<ACT, SW, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM4_LDS_BYTES));
        attr4 = true;
    }
    GemmArgs a = a0;
    a.tiles_m = (a.M + GEMM4_BM - 1) / GEMM4_BM;
    a.tiles_n = a.N / GEMM4_BN;
    hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, F32>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a);
}

// Row split: M = 1621 leaves the 256-row kernel a 7th row tile with 85 live rows (9.5 % of its MFMA work wasted) and the
// 128-row kernel is slower per FLOP.  When N is wide enough for both parts to fill the chip, the first floor(M/256)*256 rows
// go to the 256x256 kernel and the remaining rows to whatever the chooser picks for them -- two launches, same stream.  Every
// kernel accumulates K in the same order and shares one epilogue, so the output bits do not change (asserted in
// tests/test_gpu_ops.py).  Measured: 1621x28672x4096 + SwiGLU 362 -> 345 us; loses on N <= 6144 and on M % 256 > 128.
static bool want_m_split(const GemmArgs& a) {
    if (g_gemm_variant != 0 || a.out_grp > 0 || a.res_row_mod > 0 || a.N % GEMM4_BN || a.K < 2048 || a.M < 1024) return false;
    const int r = a.M % GEMM4_BM, m1_tiles = a.M / GEMM4_BM, n_tiles = a.N / GEMM4_BN;
    if (r == 0 || r > 96 || n_tiles < 64) return false;
    const long t4 = (long)m1_tiles * n_tiles;
    return (double)t4 / (double)(((t4 + 255) / 256) * 256) >= 0.85;      // the 256-row part fills its rounds
}

template <int ACT, bool SW, bool F32, bool G>
static void launch_gemm(const GemmArgs& a0, hipStream_t s) {
    if constexpr (!G && !F32) {
        if (want_m_split(a0)) {
            const int M1 = a0.M / GEMM4_BM * GEMM4_BM;
            GemmArgs hi = a0, lo = a0;
            hi.M = M1;
            launch_gemm4<ACT, SW, F32>(hi, s);
            lo.M = a0.M - M1;
            lo.A = a0.A + (size_t)M1 * a0.lda;
            lo.C = (bf16_t*)a0.C + (size_t)M1 * a0.ldc;
            if (a0.res) lo.res = a0.res + (size_t)M1 * a0.ldres;
            lo.tiles_m = (lo.M + GEMM_BM - 1) / GEMM_BM;
            launch_gemm<ACT, SW, F32, G>(lo, s);
            return;
        }
    }
    if constexpr (!SW && !F32) {
        if (choose_splitk(a0) <= 1 && want_small_m(a0)) {
            static bool attr_s = false;
            if (!attr_s) {
                check(hipFuncSetAttribute((const void*)gemm_s_bf16_kernel<ACT, G>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMMS_LDS_BYTES));
                attr_s = true;
            }
            GemmArgs a = a0;
            a.tiles_m = (a.M + GEMMS_BM - 1) / GEMMS_BM;
            a.tiles_n = a.N / GEMMS_BN;
            hipLaunchKernelGGL((gemm_s_bf16_kernel<ACT, G>), dim3(a.tiles_m * a.tiles_n), dim3(128), GEMMS_LDS_BYTES, s, a);
            return;
        }
    }
    if (const int split = choose_splitk(a0); split > 1) {
        static bool attr_k = false;
        if (!attr_k) {
            check(hipFuncSetAttribute((const void*)gemm_bf16_kernel<ACT, SW, F32, G, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GEMM_LDS_BYTES));
            attr_k = true;
        }
        GemmArgs a = a0;
        a.sk_ws = (float*)g_ws;
        a.sk_flags = (int*)((char*)g_ws + SPLITK_CNT_OFF);
        hipLaunchKernelGGL((gemm_bf16_kernel<ACT, SW, F32, G, false, true>), dim3(a.tiles_m * a.tiles_n, split), dim3(256), GEMM_LDS_BYTES, s, a);
        return;
    }
    if constexpr (!G) {
        // measured (scripts/kernel_bench.py --frames 8): wins 15-20 % at <= 256 tiles with K >= 4096, loses at K = 1024 and beyond one round
        if (g_gemm_variant == 256 || (g_gemm_variant == 0 && (long)a0.tiles_m * a0.tiles_n <= 256 && a0.K >= 4096)) {
            static bool attrl8 = false;
            if (!attrl8) {
                check(hipFuncSetAttribute((const void*)gemm_l8_bf16_kernel<ACT, SW, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMML_LDS_BYTES));
                attrl8 = true;
            }
            hipLaunchKernelGGL((gemm_l8_bf16_kernel<ACT, SW, F32>), dim3(a0.tiles_m * a0.tiles_n), dim3(512), GEMML_LDS_BYTES, s, a0);
            return;
        }
        const int kern = g_gemm_variant == 0 ? choose_gemm_kernel(a0) : g_gemm_variant;
        if (kern == 4 && a0.N % GEMM3_BN == 0) {
            static bool attr3 = false;
            if (!attr3) {
                check(hipFuncSetAttribute((const void*)gemm3_bf16_kernel<ACT, SW, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM3_LDS_BYTES));
                attr3 = true;
            }
            GemmArgs a = a0;
            a.tiles_m = (a.M + GEMM3_BM - 1) / GEMM3_BM;
            a.tiles_n = a.N / GEMM3_BN;
            hipLaunchKernelGGL((gemm3_bf16_kernel<ACT, SW, F32>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM3_LDS_BYTES, s, a);
            return;
        }
        if (kern == 8 && a0.N % GEMM4_BN == 0) {
            launch_gemm4<ACT, SW, F32>(a0, s);
            return;
        }
    }
    if constexpr (!G) {
        if (want_stream_k(a0)) {
            static bool attr_sk = false;
            if (!attr_sk) {
                check(hipFuncSetAttribute((const void*)gemm_sk_bf16_kernel<ACT, SW, F32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    GEMM_LDS_BYTES));
                attr_sk = true;
            }
            GemmArgs a = a0;
            const int total = a.tiles_m * a.tiles_n * (a.K / GEMM_BK);
            a.sk_ws = (float*)g_ws;
            a.sk_flags = (int*)((char*)g_ws + SK_FLAGS_OFF);
            a.sk_per = (total + SK_GRID - 1) / SK_GRID;
            check(hipMemsetAsync(a.sk_flags, 0, (SK_GRID + 1) * 4, s));               // flags re-armed before EVERY launch (guide G16)
            hipLaunchKernelGGL((gemm_sk_bf16_kernel<ACT, SW, F32>), dim3(SK_GRID), dim3(256), GEMM_LDS_BYTES, s, a);
            return;
        }
    }
    const GemmArgs& a = a0;
    static bool attr_set = false;   // 64 KiB dynamic LDS needs the opt-in once per kernel instance
    if (!attr_set) {
        check(hipFuncSetAttribute((const void*)gemm_bf16_kernel<ACT, SW, F32, G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            GEMM_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<ACT, SW, F32, G>), dim3(a.tiles_m * a.tiles_n), dim3(256), GEMM_LDS_BYTES, s, a);
}

extern "C" int32_t vl2_gemm_bf16(const void* A, const void* W, void* C, const float* bias, const void* res, int32_t M,
                                 int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t ldres, int32_t act,
                                 int32_t flags, const int32_t* a_idx, const void* zero_row, int32_t seg_k, int32_t out_grp,
                                 int32_t out_grp_pad, int32_t out_row_off, int32_t res_row_mod, int32_t res_row_off,
                                 void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_gemm_bf16: null pointer or empty shape");
    if (N % 128 || K % 64) return fail(VL2_E_SHAPE, "vl2_gemm_bf16: need N%%128==0 and K%%64==0 (N=%d K=%d)", N, K);
    if ((lda % 8) || (ldw % 8) || (ldc % 8) || (res && (ldres % 8)) || !ALIGNED16(A) || !ALIGNED16(W) || !ALIGNED16(C) ||
        (res && !ALIGNED16(res)) || (bias && !ALIGNED16(bias)))
        return fail(VL2_E_SHAPE, "vl2_gemm_bf16: pointers / leading dims must be 16-byte aligned");
    const bool sw = flags & VL2_GEMM_SWIGLU, f32 = flags & VL2_GEMM_OUT_F32, g = a_idx != nullptr;
    if (g && (seg_k <= 0 || seg_k % 64 || K % seg_k)) return fail(VL2_E_SHAPE, "vl2_gemm_bf16: bad gather segments");
    GemmArgs a{(const bf16_t*)A, (const bf16_t*)W, C, bias, (const bf16_t*)res, a_idx, (const bf16_t*)zero_row, M, N, K,
               lda, ldw, ldc, ldres, seg_k, out_grp, out_grp_pad, out_row_off, res_row_mod, res_row_off,
               (M + GEMM_BM - 1) / GEMM_BM, N / GEMM_BN};
    hipStream_t s = ST(stream);
    if (out_grp > 0 || res_row_mod > 0) {                      // row-remap epilogue (patch-embed): dedicated instantiation
        if (sw || g || f32 || act != VL2_ACT_NONE) return fail(VL2_E_UNSUPP, "vl2_gemm_bf16: row remap supports plain bf16 output only");
        static bool attr_r = false;
        if (!attr_r) {
            check(hipFuncSetAttribute((const void*)gemm_bf16_kernel<ACT_NONE, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
            attr_r = true;
        }
        hipLaunchKernelGGL((gemm_bf16_kernel<ACT_NONE, false, false, false, true>), dim3(a.tiles_m * a.tiles_n), dim3(256), GEMM_LDS_BYTES, s, a);
        return launched("vl2_gemm_bf16");
    }
    if (sw) {
        if (f32 || g || act != VL2_ACT_NONE || bias) return fail(VL2_E_UNSUPP, "vl2_gemm_bf16: SWIGLU excludes bias/act/f32/gather");
        launch_gemm<ACT_NONE, true, false, false>(a, s);
    } else if (g) {
        if (f32) return fail(VL2_E_UNSUPP, "vl2_gemm_bf16: gather + f32 not built");
        if (act == VL2_ACT_SILU) launch_gemm<ACT_SILU, false, false, true>(a, s);
        else if (act == VL2_ACT_NONE) launch_gemm<ACT_NONE, false, false, true>(a, s);
        else return fail(VL2_E_UNSUPP, "vl2_gemm_bf16: gather supports act none/silu");
    } else if (f32) {
        if (act != VL2_ACT_NONE) return fail(VL2_E_UNSUPP, "vl2_gemm_bf16: f32 output supports act none");
        launch_gemm<ACT_NONE, false, true, false>(a, s);
    } else {
        switch (act) {
            case VL2_ACT_NONE: launch_gemm<ACT_NONE, false, false, false>(a, s); break;
            case VL2_ACT_QGELU: launch_gemm<ACT_QGELU, false, false, false>(a, s); break;
            case VL2_ACT_GELU: launch_gemm<ACT_GELU, false, false, false>(a, s); break;
            case VL2_ACT_SILU: launch_gemm<ACT_SILU, false, false, false>(a, s); break;
            case VL2_ACT_GELU_TANH: launch_gemm<ACT_GELU_TANH, false, false, false>(a, s); break;
            default: return fail(VL2_E_UNSUPP, "vl2_gemm_bf16: unknown act %d", act);
        }
    }
    return launched("vl2_gemm_bf16");
}

// ------------------------------------------------------------------------------------------------ skinny-M GEMM (batched decode)
template <int MT>
static void launch_skinny(const SkinnyArgs& a, int ks, size_t lds, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        check(hipFuncSetAttribute((const void*)gemm_skinny_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<MT>), dim3(a.N / 64, ks), dim3(256), lds, s, a);
}
extern "C" int32_t vl2_gemm_skinny_bf16(const void* A, const void* W, void* C, const float* bias, const void* res, int32_t M,
                                        int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t ldres, int32_t flags,
                                        void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_gemm_skinny_bf16: null pointer or empty shape");
    const bool sw = flags & VL2_GEMM_SWIGLU, f32 = flags & VL2_GEMM_OUT_F32;
    if (M > 64 || N % 64 || (sw && N % 128) || K % 32 || lda % 8 || ldw % 8 || ldc % 4 || (res && ldres % 4))
        return fail(VL2_E_SHAPE, "vl2_gemm_skinny_bf16: need M<=64, N%%64==0, K%%32==0 (M=%d N=%d K=%d)", M, N, K);
    if (sw && (bias || f32)) return fail(VL2_E_UNSUPP, "vl2_gemm_skinny_bf16: SWIGLU excludes bias / f32 output");
    if (!g_ws) return fail(VL2_E_BADARG, "vl2_gemm_skinny_bf16: needs vl2_set_workspace (fp32 partial sums)");
    const int mt = M <= 16 ? 1 : M <= 32 ? 2 : 4, Mp = 16 * mt;
    const int steps = K / 32;
    // K split: enough (column group, K slice) waves to keep ~4096 in flight, a divisor of the 32-deep steps, partials within
    // the workspace
    int ks = (4096 + N / 16 - 1) / (N / 16);
    ks = ks < 1 ? 1 : ks > 32 ? 32 : ks;
    while (ks > 1 && (steps % ks || (int64_t)ks * Mp * N * 4 > g_ws_bytes)) --ks;
    const int kslice = K / ks;
    int kchunk = kslice;                                      // largest 32-multiple divisor of the slice whose x chunk fits 64 KiB
    while (kchunk > 32 && (kslice % kchunk || kchunk % 32 || (size_t)Mp * (kchunk + 8) * 2 > 65536)) kchunk -= 32;
    if (kslice % kchunk || (size_t)Mp * (kchunk + 8) * 2 > 65536) return fail(VL2_E_SHAPE, "vl2_gemm_skinny_bf16: no K chunking for K=%d", K);
    SkinnyArgs a{(const bf16_t*)A, (const bf16_t*)W, (float*)g_ws, M, N, K, lda, ldw, kslice, kchunk};
    const size_t lds = (size_t)Mp * (kchunk + 8) * 2;
    hipStream_t s = ST(stream);
    if (mt == 1) launch_skinny<1>(a, ks, lds, s); else if (mt == 2) launch_skinny<2>(a, ks, lds, s); else launch_skinny<4>(a, ks, lds, s);
    SkinnyReduceArgs r{(const float*)g_ws, C, bias, (const bf16_t*)res, M, Mp, N, ks, ldc, ldres};
    const int ncol = sw ? N / 2 : N;
    const dim3 g((M * (ncol / 4) + 255) / 256), b(256);
    if (sw) hipLaunchKernelGGL((skinny_reduce_kernel<true, false>), g, b, 0, s, r);
    else if (f32) hipLaunchKernelGGL((skinny_reduce_kernel<false, true>), g, b, 0, s, r);
    else hipLaunchKernelGGL((skinny_reduce_kernel<false, false>), g, b, 0, s, r);
    return launched("vl2_gemm_skinny_bf16");
}

// ------------------------------------------------------------------------------------------------ norms
static int32_t launch_norm(const NormArgs& a, bool rms, hipStream_t s, const char* what) {
    if (!a.x || !a.y || !a.w || a.rows <= 0 || a.C <= 0) return fail(VL2_E_BADARG, "%s: null pointer or empty shape", what);
    if (a.C % 8 || a.C > 8192 || a.ldx % 8 || a.ldy % 8 || (a.res && a.ldres % 8))
        return fail(VL2_E_SHAPE, "%s: need C%%8==0, C<=8192, aligned strides (C=%d)", what, a.C);
    const int nv = (a.C + 511) / 512;
    dim3 g((a.rows + 3) / 4), b(256);
    if (rms && a.C > 2048) {     // long rows: the wide form (a workgroup per row), whatever the row count -- a row's result must
                                 // not depend on how many rows are normalised with it (batched prefill == one by one)
        if (a.C <= 4096) hipLaunchKernelGGL((norm_wide_kernel<true, 2>), dim3(a.rows), b, 0, s, a);
        else hipLaunchKernelGGL((norm_wide_kernel<true, 4>), dim3(a.rows), b, 0, s, a);
    } else if (rms) {
        if (nv <= 1) hipLaunchKernelGGL((norm_kernel<1, true>), g, b, 0, s, a);
        else if (nv <= 2) hipLaunchKernelGGL((norm_kernel<2, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((norm_kernel<8, true>), g, b, 0, s, a);
    } else {
        if (nv <= 1) hipLaunchKernelGGL((norm_kernel<1, false>), g, b, 0, s, a);
        else if (nv <= 2) hipLaunchKernelGGL((norm_kernel<2, false>), g, b, 0, s, a);
        else if (nv <= 3) hipLaunchKernelGGL((norm_kernel<3, false>), g, b, 0, s, a);      // SigLIP hidden 1152
        else if (nv <= 8) hipLaunchKernelGGL((norm_kernel<8, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((norm_kernel<16, false>), g, b, 0, s, a);                  // STC at the 72B decoder's width 8192
    }
    return launched(what);
}
extern "C" int32_t vl2_layernorm(const void* x, void* y, const float* w, const float* b, const void* res, int32_t rows,
                                 int32_t C, int32_t ldx, int32_t ldy, int32_t ldres, float eps, int32_t silu, void* stream) {
    NormArgs a{(const bf16_t*)x, (bf16_t*)y, w, b, (const bf16_t*)res, rows, C, ldx, ldy, ldres, eps, silu};
    return launch_norm(a, false, ST(stream), "vl2_layernorm");
}
extern "C" int32_t vl2_rmsnorm(const void* x, void* y, const float* w, int32_t rows, int32_t C, int32_t ldx, int32_t ldy,
                               float eps, void* stream) {
    NormArgs a{(const bf16_t*)x, (bf16_t*)y, w, nullptr, nullptr, rows, C, ldx, ldy, 0, eps, 0};
    return launch_norm(a, true, ST(stream), "vl2_rmsnorm");
}

// ------------------------------------------------------------------------------------------------ ViT front end
extern "C" int32_t vl2_patchify(const void* frames, int32_t dtype, void* out, int32_t T, int32_t H, int32_t W, int32_t P,
                                int32_t G, int32_t Kp, void* stream) {
    if (!frames || !out || T <= 0) return fail(VL2_E_BADARG, "vl2_patchify: null pointer or empty shape");
    if (G * P > H || G * P > W || Kp % 8 || Kp < 3 * P * P) return fail(VL2_E_SHAPE, "vl2_patchify: bad geometry");
    dim3 g(G, T), b(256);
    hipStream_t s = ST(stream);
    if (dtype == 0) hipLaunchKernelGGL((patchify_kernel<float>), g, b, 0, s, (const float*)frames, (bf16_t*)out, H, W, P, G, Kp);
    else if (dtype == 1) hipLaunchKernelGGL((patchify_kernel<_Float16>), g, b, 0, s, (const _Float16*)frames, (bf16_t*)out, H, W, P, G, Kp);
    else if (dtype == 2) hipLaunchKernelGGL((patchify_kernel<bf16_t>), g, b, 0, s, (const bf16_t*)frames, (bf16_t*)out, H, W, P, G, Kp);
    else return fail(VL2_E_UNSUPP, "vl2_patchify: dtype %d", dtype);
    return launched("vl2_patchify");
}
extern "C" int32_t vl2_patchify_u8(const void* frames_thwc, void* out, int32_t T, int32_t H, int32_t W, int32_t P, int32_t G, int32_t Kp,
                                   float rescale, const float* mean3, const float* std3, void* stream) {
    if (!frames_thwc || !out || !mean3 || !std3 || T <= 0) return fail(VL2_E_BADARG, "vl2_patchify_u8: null pointer or empty shape");
    if (G * P > H || G * P > W || Kp % 8 || Kp < 3 * P * P) return fail(VL2_E_SHAPE, "vl2_patchify_u8: bad geometry");
    U8Norm n{rescale, {mean3[0], mean3[1], mean3[2]}, {1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2]}};   // host pointers
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(G, T), dim3(256), 0, ST(stream), (const unsigned char*)frames_thwc, (bf16_t*)out, H, W, P, G, Kp, n);
    return launched("vl2_patchify_u8");
}
extern "C" int32_t vl2_fill_cls(void* x, const void* cls_pos, int32_t T, int32_t D, int32_t rows_per_frame, void* stream) {
    if (!x || !cls_pos || T <= 0 || D % 8) return fail(VL2_E_BADARG, "vl2_fill_cls: bad args");
    hipLaunchKernelGGL(fill_cls_kernel, dim3(T), dim3(128), 0, ST(stream), (bf16_t*)x, (const bf16_t*)cls_pos, D, rows_per_frame);
    return launched("vl2_fill_cls");
}

// ------------------------------------------------------------------------------------------------ attention
extern "C" int32_t vl2_attn_fwd(const void* q, const void* k, const void* v, void* o, int64_t q_bs, int64_t q_hs, int32_t q_rs,
                                int64_t k_bs, int64_t k_hs, int32_t k_rs, int64_t v_bs, int64_t v_hs, int32_t v_rs,
                                int64_t o_bs, int64_t o_hs, int32_t o_rs, int32_t B, int32_t H, int32_t nq, int32_t nk,
                                int32_t group, float scale, int32_t causal, int32_t causal_off, int32_t D, void* stream) {
    if (!q || !k || !v || !o || B <= 0 || H <= 0 || nq <= 0 || nk <= 0 || group <= 0)
        return fail(VL2_E_BADARG, "vl2_attn_fwd: null pointer or empty shape");
    if ((q_rs | k_rs | v_rs | o_rs) % 8 || (q_bs | q_hs | k_bs | k_hs | v_bs | v_hs | o_bs | o_hs) % 4 || !ALIGNED16(q) ||
        !ALIGNED16(k) || !ALIGNED16(v) || ((uintptr_t)o & 7))
        return fail(VL2_E_SHAPE, "vl2_attn_fwd: strides must keep 16-byte row alignment");
    if (causal && causal_off < 0) return fail(VL2_E_SHAPE, "vl2_attn_fwd: causal_off must be >= 0");
    // K / V tiles are fetched through raw buffer resources whose byte offsets and NUM_RECORDS are 32-bit
    if (((int64_t)(nk - 1) * k_rs + D) * 2 >= (int64_t)1 << 31 || ((int64_t)(nk - 1) * v_rs + D) * 2 >= (int64_t)1 << 31)
        return fail(VL2_E_SHAPE, "vl2_attn_fwd: one head's K or V rows span >= 2 GiB (nk %d, row strides %d / %d elements)", nk, k_rs, v_rs);
    AttnArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, q_bs, q_hs, q_rs, k_bs, k_hs, k_rs,
               v_bs, v_hs, v_rs, o_bs, o_hs, o_rs, nq, nk, group, H, B, scale * 1.4426950408889634f, causal_off};
    dim3 g((nq + 127) / 128, H, B), b(256);
    if (causal) g = dim3(((nq + 127) / 128) * H * B, 1, 1);
    hipStream_t s = ST(stream);
    if (D == 64 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<64, false>), g, b, 0, s, a);
    else if (D == 64 && causal) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), g, b, 0, s, a);
    else if (D == 128 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<128, false>), g, b, 0, s, a);
    else if (D == 128 && causal) {
        // two KV groups per workgroup when a sequence has few (q block, head) pairs: decided per SEQUENCE (B is left out) so
        // that a prompt gets the same bits prefilled alone or in a batch.  Measured (scripts/attn_bench.py): 256 pairs (S = 945,
        // 32 heads) 28.5 -> 27.1 us, 336 (S = 1452, 28 heads) 38.9 -> 35.8; 416 (S = 1621, 32 heads) 47.4 -> 52.2: the SIMD's
        // per-tile throughput, not the length of the dependent tile chain, is the limit once every CU has > 1.4 workgroups.
        const long per_seq = (long)((nq + 127) / 128) * H;
        const bool two = g_attn_kv_groups == 2 || (g_attn_kv_groups == 0 && per_seq <= 352);
        if (two) hipLaunchKernelGGL((attn_fwd_kernel<128, true, 2>), g, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<128, true>), g, b, 0, s, a);
    }
    else if (D == 96 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<96, false>), g, b, 0, s, a);
    else return fail(VL2_E_SHAPE, "vl2_attn_fwd: head_dim %d not built (64, 96 non-causal, 128)", D);
    return launched("vl2_attn_fwd");
}

// ------------------------------------------------------------------------------------------------ STC direct kernels
extern "C" int32_t vl2_dwconv3x3_ln_silu(const void* x, void* y, const float* w9c, const float* lnw, const float* lnb, int32_t F,
                                         int32_t H, int32_t W, int32_t C, float eps, void* stream) {
    if (!x || !y || !w9c || !lnw || !lnb || F <= 0 || H <= 0 || W <= 0) return fail(VL2_E_BADARG, "vl2_dwconv3x3_ln_silu: bad args");
    if (C % 8 || C > 8192) return fail(VL2_E_SHAPE, "vl2_dwconv3x3_ln_silu: need C%%8==0 and C<=8192");
    dim3 g(F * H * W), b(256);
    if (W >= 16) {         // wide rows: four positions per workgroup (k_stc.h); the choice depends on W alone
        dim3 g4(F * H * ((W + DW_P - 1) / DW_P));
        if (C <= 2048) hipLaunchKernelGGL((dwconv4_ln_silu_kernel<1>), g4, b, 0, ST(stream), (const bf16_t*)x, (bf16_t*)y, w9c, lnw, lnb, H, W, C, eps);
        else if (C <= 4096) hipLaunchKernelGGL((dwconv4_ln_silu_kernel<2>), g4, b, 0, ST(stream), (const bf16_t*)x, (bf16_t*)y, w9c, lnw, lnb, H, W, C, eps);
        else hipLaunchKernelGGL((dwconv4_ln_silu_kernel<4>), g4, b, 0, ST(stream), (const bf16_t*)x, (bf16_t*)y, w9c, lnw, lnb, H, W, C, eps);
        return launched("vl2_dwconv3x3_ln_silu");
    }
    if (C <= 2048) hipLaunchKernelGGL((dwconv_ln_silu_kernel<1>), g, b, 0, ST(stream), (const bf16_t*)x, (bf16_t*)y, w9c, lnw, lnb, H, W, C, eps);
    else if (C <= 4096) hipLaunchKernelGGL((dwconv_ln_silu_kernel<2>), g, b, 0, ST(stream), (const bf16_t*)x, (bf16_t*)y, w9c, lnw, lnb, H, W, C, eps);
    else hipLaunchKernelGGL((dwconv_ln_silu_kernel<4>), g, b, 0, ST(stream), (const bf16_t*)x, (bf16_t*)y, w9c, lnw, lnb, H, W, C, eps);
    return launched("vl2_dwconv3x3_ln_silu");
}
extern "C" int32_t vl2_chan_mean(const void* x, float* mean, int32_t F, int32_t HW, int32_t C, void* stream) {
    if (!x || !mean || F <= 0 || HW <= 0) return fail(VL2_E_BADARG, "vl2_chan_mean: bad args");
    if (C % 64) return fail(VL2_E_SHAPE, "vl2_chan_mean: need C%%64==0");
    hipLaunchKernelGGL(chan_mean_kernel, dim3(C / 64, F), dim3(256), 0, ST(stream), (const bf16_t*)x, mean, HW, C);
    return launched("vl2_chan_mean");
}
extern "C" int32_t vl2_small_linear(const float* x, const void* W, const float* b, float* out, int32_t F, int32_t N, int32_t K,
                                    int32_t act, void* stream) {
    if (!x || !W || !out || F <= 0 || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_small_linear: bad args");
    if (K % 8) return fail(VL2_E_SHAPE, "vl2_small_linear: need K%%8==0");
    const int a = act == VL2_ACT_SILU ? 1 : act == VL2_ACT_SIGMOID ? 2 : act == VL2_ACT_NONE ? 0 : -1;
    if (a < 0) return fail(VL2_E_UNSUPP, "vl2_small_linear: act %d", act);
    hipLaunchKernelGGL(small_linear_kernel, dim3((N + SL_NB - 1) / SL_NB), dim3(256), 0, ST(stream), x, (const bf16_t*)W, b, out, F, N, K, a);
    return launched("vl2_small_linear");
}
extern "C" int32_t vl2_se_scale(void* x, const float* gate, int32_t F, int32_t HW, int32_t C, void* stream) {
    if (!x || !gate || F <= 0 || HW <= 0 || C % 8) return fail(VL2_E_BADARG, "vl2_se_scale: bad args");
    const size_t nvec = (size_t)F * HW * C / 8;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), (bf16_t*)x, gate, HW, C, nvec);
    return launched("vl2_se_scale");
}

// ------------------------------------------------------------------------------------------------ decoder glue / decode
extern "C" int32_t vl2_rope_kv(const void* qkv, void* q_out, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                               int32_t S, int32_t nh, int32_t nkv, int32_t smax, int32_t pos0, void* stream) {
    if (!qkv || !q_out || !kcache || !vcache || !cos_t || !sin_t || S <= 0) return fail(VL2_E_BADARG, "vl2_rope_kv: bad args");
    if (pos0 < 0 || pos0 + S > smax) return fail(VL2_E_SHAPE, "vl2_rope_kv: positions %d..%d exceed the cache (%d)", pos0, pos0 + S, smax);
    const size_t total = (size_t)S * (nh + 2 * nkv) * 8;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rope_kv_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), (const bf16_t*)qkv, (bf16_t*)q_out,
                       (bf16_t*)kcache, (bf16_t*)vcache, cos_t, sin_t, S, nh, nkv, smax, pos0);
    return launched("vl2_rope_kv");
}

template <bool SW, bool F32>
static void launch_gemv(const GemvArgs& a, int n_out, hipStream_t s) {
    if (g_gemv_rpw == 2)
        hipLaunchKernelGGL((gemv_bf16_kernel<SW, F32, 2>), dim3((n_out + 7) / 8), dim3(256), (size_t)a.K * 2, s, a);
    else if (g_gemv_rpw == 4)
        hipLaunchKernelGGL((gemv_bf16_kernel<SW, F32, 4>), dim3((n_out + 15) / 16), dim3(256), (size_t)a.K * 2, s, a);
    else
        hipLaunchKernelGGL((gemv_bf16_kernel<SW, F32, 1>), dim3((n_out + 3) / 4), dim3(256), (size_t)a.K * 2, s, a);
}
extern "C" int32_t vl2_gemv_bf16(const void* W, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                                 int32_t N, int32_t K, int32_t ldw, float eps, int32_t flags, void* stream) {
    if (!W || !x || !y || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_gemv_bf16: bad args");
    if (K % 8 || ldw % 8 || K > 32704) return fail(VL2_E_SHAPE, "vl2_gemv_bf16: need K%%8==0, K<=32704 (x lives in LDS; K=%d)", K);
    const bool sw = flags & VL2_GEMM_SWIGLU, f32 = flags & VL2_GEMM_OUT_F32;
    if (sw && (N % 64 || f32 || bias)) return fail(VL2_E_SHAPE, "vl2_gemv_bf16: SWIGLU needs N%%64==0, bf16 output, no bias");
    GemvArgs a{(const bf16_t*)W, (const bf16_t*)x, norm_w, (const bf16_t*)res, y, N, K, ldw, eps, bias, 0, 0, 0};
    if (sw) launch_gemv<true, false>(a, N / 2, ST(stream));
    else if (f32) launch_gemv<false, true>(a, N, ST(stream));
    else launch_gemv<false, false>(a, N, ST(stream));
    return launched("vl2_gemv_bf16");
}
template <bool SW, bool F32>
static void launch_gemv_mr(const GemvArgs& a, int mb, int n_out, hipStream_t s) {
    const int rpw = g_gemv_mr_rpw;                    // output rows per wave (amortises staging MB rows of x)
    const dim3 g((n_out + 4 * rpw - 1) / (4 * rpw)), b(256);
    const size_t lds = (size_t)mb * a.K * 2;
#define VL2_MR(MBV) do { if (rpw == 1) hipLaunchKernelGGL((gemv_mr_bf16_kernel<SW, F32, MBV, 1>), g, b, lds, s, a); \
                         else if (rpw == 2) hipLaunchKernelGGL((gemv_mr_bf16_kernel<SW, F32, MBV, 2>), g, b, lds, s, a); \
                         else hipLaunchKernelGGL((gemv_mr_bf16_kernel<SW, F32, MBV, 4>), g, b, lds, s, a); } while (0)
    if (mb == 2) VL2_MR(2); else if (mb == 3) VL2_MR(3); else VL2_MR(4);
#undef VL2_MR
}
extern "C" int32_t vl2_gemv_batched_bf16(const void* W, const void* x, const float* norm_w, const void* res, const float* bias,
                                         void* y, int32_t MB, int32_t N, int32_t K, int32_t ldw, int32_t ldx, int32_t ldy,
                                         int32_t ldres, float eps, int32_t flags, void* stream) {
    if (!W || !x || !y || N <= 0 || K <= 0 || MB <= 0) return fail(VL2_E_BADARG, "vl2_gemv_batched_bf16: bad args");
    if (K % 8 || ldw % 8 || ldx % 8 || K > 32704) return fail(VL2_E_SHAPE, "vl2_gemv_batched_bf16: need K%%8==0, K<=32704 (K=%d)", K);
    const bool sw = flags & VL2_GEMM_SWIGLU, f32 = flags & VL2_GEMM_OUT_F32;
    if (sw && (N % 64 || f32 || bias)) return fail(VL2_E_SHAPE, "vl2_gemv_batched_bf16: SWIGLU needs N%%64==0, bf16 output, no bias");
    const int n_out = sw ? N / 2 : N;
    const int esz = f32 ? 4 : 2;
    const int cap = 65536 / (K * 2) < 4 ? 65536 / (K * 2) : 4;        // x rows that fit the 64 KiB of LDS, at most 4 per pass
    for (int b0 = 0; b0 < MB;) {
        const int mb = MB - b0 < cap ? MB - b0 : cap;
        GemvArgs a{(const bf16_t*)W, (const bf16_t*)x + (size_t)b0 * ldx, norm_w, res ? (const bf16_t*)res + (size_t)b0 * ldres : nullptr,
                   (char*)y + (size_t)b0 * ldy * esz, N, K, ldw, eps, bias, ldx, ldy, ldres};
        if (mb == 1) {
            if (sw) launch_gemv<true, false>(a, n_out, ST(stream));
            else if (f32) launch_gemv<false, true>(a, n_out, ST(stream));
            else launch_gemv<false, false>(a, n_out, ST(stream));
        } else {
            if (sw) launch_gemv_mr<true, false>(a, mb, n_out, ST(stream));
            else if (f32) launch_gemv_mr<false, true>(a, mb, n_out, ST(stream));
            else launch_gemv_mr<false, false>(a, mb, n_out, ST(stream));
        }
        b0 += mb;
    }
    return launched("vl2_gemv_batched_bf16");
}
extern "C" int32_t vl2_attn_decode(const void* qkv, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                                   float* partial, void* out, int32_t nh, int32_t nkv, int32_t smax, int32_t pos,
                                   const int32_t* pos_dev, int32_t ctx_cap, float scale, void* stream) {
    if (!qkv || !kcache || !vcache || !cos_t || !sin_t || !partial || !out || nh <= 0 || nkv <= 0)
        return fail(VL2_E_BADARG, "vl2_attn_decode: bad args");
    const int group = nh / nkv;
    if (group * nkv != nh) return fail(VL2_E_SHAPE, "vl2_attn_decode: need nh = nkv*group");
    const int cap = pos_dev ? ctx_cap : pos + 1;                 // positions the launch must be able to cover
    if (cap <= 0 || cap > smax || (!pos_dev && pos < 0)) return fail(VL2_E_SHAPE, "vl2_attn_decode: position %d outside the cache (%d)", cap - 1, smax);
    const int nsplit = (cap + 63) / 64;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(nsplit, nkv, (group + 3) / 4), dim3(256), 0, ST(stream), (const bf16_t*)qkv, (bf16_t*)kcache,
                       (bf16_t*)vcache, cos_t, sin_t, partial, nh, group, nkv, smax, pos, pos_dev, scale * 1.4426950408889634f, 0L, 0L, 0L);
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(nh), dim3(128), 0, ST(stream), partial, (bf16_t*)out, nsplit, pos, pos_dev, 0L, 0L);
    return launched("vl2_attn_decode");
}
extern "C" int32_t vl2_attn_decode_batched(const void* qkv, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                                           float* partial, void* out, int32_t B, int64_t qkv_bs, int64_t cache_bs, int64_t out_bs,
                                           int32_t nh, int32_t nkv, int32_t smax, const int32_t* pos_dev, int32_t ctx_cap, float scale,
                                           void* stream) {
    if (!qkv || !kcache || !vcache || !cos_t || !sin_t || !partial || !out || !pos_dev || nh <= 0 || nkv <= 0 || B <= 0)
        return fail(VL2_E_BADARG, "vl2_attn_decode_batched: bad args");
    const int group = nh / nkv;
    if (group * nkv != nh) return fail(VL2_E_SHAPE, "vl2_attn_decode_batched: need nh = nkv*group");
    if (ctx_cap <= 0 || ctx_cap > smax) return fail(VL2_E_SHAPE, "vl2_attn_decode_batched: ctx_cap %d outside the cache (%d)", ctx_cap, smax);
    const int nsplit = (ctx_cap + 63) / 64;
    const long partial_bs = (long)nh * nsplit * 130;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(nsplit, nkv * B, (group + 3) / 4), dim3(256), 0, ST(stream), (const bf16_t*)qkv,
                       (bf16_t*)kcache, (bf16_t*)vcache, cos_t, sin_t, partial, nh, group, nkv, smax, 0, pos_dev,
                       scale * 1.4426950408889634f, (long)qkv_bs, (long)cache_bs, partial_bs);
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(nh, B), dim3(128), 0, ST(stream), partial, (bf16_t*)out, nsplit, 0, pos_dev,
                       partial_bs, (long)out_bs);
    return launched("vl2_attn_decode_batched");
}
extern "C" int32_t vl2_argmax(const float* logits, int32_t V, int32_t* tok, int32_t* hist, int32_t step, int32_t* state,
                              void* stream) {
    if (!logits || !tok || V <= 0) return fail(VL2_E_BADARG, "vl2_argmax: bad args");
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, ST(stream), logits, V, tok, hist, step, state);
    return launched("vl2_argmax");
}
extern "C" int32_t vl2_embed_rows(const int32_t* ids, const void* table, void* out, int32_t n, int32_t D, int32_t ldo, void* stream) {
    if (!ids || !table || !out || n <= 0 || D % 8 || ldo % 8) return fail(VL2_E_BADARG, "vl2_embed_rows: bad args");
    hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(128), 0, ST(stream), ids, (const bf16_t*)table, (bf16_t*)out, D, ldo);
    return launched("vl2_embed_rows");
}

// libvl2hip.so: extern "C" launchers (include/vl2hip.h) over the gfx950 kernels in k_*.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC vl2_abi.hip -o libvl2hip.so   (see build.py)
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>

#define VL2_EXPERIMENTAL 1      // this translation unit DEFINES the experimental entry points too
#include "../../include/vl2hip.h"
#include "k_attn.h"
#include "k_attn2.h"
#include "k_decode.h"
#include "k_fp8.h"
#include "k_gemm.h"
#include "k_gemm6.h"
#include "k_gemm7.h"
#include "k_gemm9.h"
// libvl2hip.so = the default path + the documented options.  The experiments that were measured and lost (the four-wave 256x256 tile, the
// issue-order / stamped forms of the 16x16x32 kernel, the woven LDS-DMA issue outside the 192-row tiles, the two-accumulator persistent
// form, stream-K, the decode tail engine, attention + elected combine in one launch) are compiled only with -DVL2_LAB into
// libvl2hip_lab.so (scripts/build_lab_lib.sh); in the product build a request for one of them is refused with VL2_E_UNSUPP.
#ifdef VL2_LAB
#include "k_decode_tail.h"
#include "k_gemm8.h"
static constexpr bool kLab = true;
#else
static constexpr bool kLab = false;
#endif
#include "k_norm.h"
#include "k_pack.h"
#include "k_sample.h"
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
static thread_local hipError_t g_setup_err = hipSuccess;
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

// Dynamic-LDS opt-in of a kernel instance (> 64 KiB needs hipFuncSetAttribute once per device).  Thread-safe and idempotent:
// one atomic bit per device ordinal; two threads racing on the first launch both set the same attribute.
template <auto Kern>
static void lds_attr(int bytes) {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    check(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        check(hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done.fetch_or(bit, std::memory_order_release);
    }
}

extern "C" int32_t vl2_version(void) { return VL2_ABI_VERSION; }
extern "C" const char* vl2_elem_name(void) { return VL2_ELEM_NAME; }
extern "C" const char* vl2_last_error_string(void) { return g_err; }

#define SK_GRID 512                  // persistent stream-K workgroups: 2 per CU
#define SPLITK_MAX_WG 1024           // split-K: at most this many (tile, split) workgroups -> 64 MiB of fp32 partials
#define SPLITK_MAX_TILES 192         // split-K only when the plain grid leaves most of the 512 resident slots empty
// workspace layout: [SPLITK_MAX_WG][64][256] fp32 partial tiles (stream-K uses the first SK_GRID) | stream-K flags
// [SK_GRID + 1] | split-K tile counters [SPLITK_MAX_TILES] (zero when allocated, re-armed by the kernel itself)
#define SK_FLAGS_OFF ((int64_t)SPLITK_MAX_WG * 64 * 256 * 4)
#define SPLITK_CNT_OFF (SK_FLAGS_OFF + (int64_t)(SK_GRID + 1) * 4 + 12)
#define GEMM6_CTR_OFF ((SPLITK_CNT_OFF + (int64_t)SPLITK_MAX_TILES * 4 + 15) / 16 * 16)   // persistent GEMM: {tiles handed out, workgroups finished}
#define SK_WS_BYTES (GEMM6_CTR_OFF + 16)
extern "C" int64_t vl2_workspace_bytes(void) { return SK_WS_BYTES; }

__global__ void fill_zero_kernel(uint32_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
extern "C" int32_t vl2_fill_zero(void* p, int64_t bytes, void* stream) {
    if (!p || bytes < 0 || (bytes & 3) || ((uintptr_t)p & 3)) return fail(VL2_E_BADARG, "vl2_fill_zero: null / unaligned pointer or size");
    if (bytes == 0) return 0;
    const int64_t n = bytes / 4;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid), dim3(256), 0, ST(stream), (uint32_t*)p, n);
    return launched("vl2_fill_zero");
}

// per-call launch controls (vl2_gemm_desc: ws / ws_bytes / variant / VL2_GEMM_SPLITK) -- nothing of this is process state
struct GemmCtl {
    void* ws;
    int64_t ws_bytes;
    int variant;
    bool splitk;
    bool no_persist = false;       // variant 24: the automatic choice without the persistent form
    bool persist = false;          // VL2_GEMM_PERSISTENT: the automatic choice may take the persistent form
    bool no_mix = false;           // VL2_GEMM_NO_MIX
    bool no_fill = false;          // VL2_GEMM_NO_FILL
    bool weave = false;            // VL2_GEMM_WEAVE
    bool no_weave4 = false;        // VL2_GEMM_NO_WEAVE4: the 192-row tiles with the load-phase issue of rounds 3-4 (A/B)
    bool mfma16 = false;           // VL2_GEMM_MFMA16 / variant 16: the 256 x 256 ping-pong kernel on v_mfma_f32_16x16x32_bf16 (k_gemm9.h; other bits)
    bool weave4 = false;           // VL2_GEMM_WEAVE4: the 256 x 256 / 192 x 256 ping-pong bodies issue their LDS-DMA from the matrix phases (k_gemm.h gemm4_body WEAVE4)
    bool* fin = nullptr;           // set to true by a launch path whose kernel has no producer-side finalize (gemm_rows_ticket): vl2_gemm then
                                   // appends the row_norm_finalize launch itself, so `row_norm_out` is filled whichever kernel ran
};

// ------------------------------------------------------------------------------------------------ GEMM
// stream-K form: measured 0.5-0.6x of the plain grid on this workload's shapes (per-tile prologue/epilogue of the persistent
// workgroups, ~15 us of partial-tile exchange, worse L2 locality of strided tile ownership) -> only on explicit request.
static bool want_stream_k(const GemmArgs& a, const GemmCtl& c) { return c.ws && c.variant == 2 && !a.norm && !a.stats_out; }

// Tile-shape choice (auto): expected efficiency = how full the last round of resident workgroups is x rows wasted by the
// M edge x the kernel's measured rate on well-quantised shapes (128x128 two-barrier kernel 1.0, 128x256 ping-pong 1.07,
// 256x256 ping-pong 1.2: profiles/r01_gemm_experiments.md).  Fitted to the measured shapes of the T=16 workload: the LLM
// o/gate-up/down projections and the STC 4096x4096 convs take 128x256, ViT qkv/wo/fc2 and the LLM qkv take 256x256,
// short-K GEMMs (STC b1) stay on 128x128.  Returns 1, 4 (gemm3), 8 (gemm4) or 12 (gemm4 on 192-row tiles).
static int choose_gemm_kernel(const GemmArgs& a, double* eff = nullptr) {
    if (eff) *eff = 0.0;
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
    // the 256x256 kernel on 192-row tiles (variant 12): the same rate per FLOP over whole rounds (sq 4096^3: 85 us per round of 192-row
    // tiles against 117), a little less in practice (fc1 at 3.06 rounds 105.9 us vs 95.4 on 256 rows) -> 1.13.  Takes the shapes whose
    // 256-row grid leaves a round badly filled: ViT out_proj / fc2 at 16 frames (148 -> 196 tiles: 30.7 -> 28.9 us, 84.6 -> 81.3), the
    // LLM q/k/v at S = 1621 (168 -> 216 tiles: 84.0 -> 81.2), the STC K = 1024 conv (576 = 2.25 rounds -> 768 = 3.0: 87.6 -> 82.2).
    // scripts/ubench/gemm_lab.hip (bit-identical, hash-checked) and scripts/kernel_bench.py, profiles/r03_gemm_lab_t192.txt
    const double m192 = (double)a.M / (((a.M + 191) / 192) * 192.0);
    const double e5 = fill((double)((a.M + 191) / 192) * (a.N / 256) / 256.0) * m192 * 1.13;
    if (a.K >= 512 && e5 > eb * 1.02) { best = 12; eb = e5; }
    // ... and on 160-row tiles (variant 10, round 6: the 192-row tile without group 1's third row block): ONE-round grids only, where a finer tile
    // shortens the single round -- ViT out_proj / fc2 at 16 frames: 196 tiles of 192 rows (77 % of the CUs) -> 232 tiles of 160 rows (91 %).  Rate 1.08:
    // a SIMD issues 20 MFMAs per slab behind the same load phases (24 on 192 rows).
    // MEASURED AND NOT TAKEN (profiles/r06_tile160_bench.txt, interleaved on one box): fc2 9232 x 1024 x 4096 79.2 us on 192 rows, 82.7 on 160; out_proj 30.4 vs
    // 30.9; 18464 rows: 163.5 vs 166.5.  A one-round grid at 77 % of the CUs is NOT 77 % of the chip: the part runs at its power limit, and 196 workgroups at a
    // higher clock do what 232 do at a lower one -- filling the round buys nothing where watts, not CUs, are the budget.  The tile stays a lab form (variant 10).
    if (eff) *eff = best == 1 ? e1 : eb;
    return best;
}

// gemm7 (k_gemm7.h): 224 x 128 / 192 x 128 tiles for GEMMs that are ONE round of workgroups whatever the tile -- 0 = not this call, else R1
// (3 = 224 rows, 2 = 192 rows).  Same efficiency model as choose_gemm_kernel: fill of the last round x rows wasted at the M edge x the
// kernel's rate relative to the 128x128 kernel (GEMM7_RATE, measured: profiles/r05_experiments.md).  Only where the 128x128 grid is
// more than one tile per CU (below that the one-round 128x128 / 64x64 kernels own the shape) and the K loop is long enough to pay for
// the image epilogue.
#define GEMM7_RATE 0.88
static int choose_gemm7(const GemmArgs& a, bool gather) {
    if (a.N % GEMM7_BN || a.K < 1024 || a.out_grp > 0 || a.res_row_mod > 0) return 0;
    const long t128 = (long)((a.M + 127) / 128) * (a.N / 128);
    if (t128 <= 256) return 0;
    const auto fill = [](double rounds) { return rounds / (double)(long)(rounds + 0.999999); };
    double eb = 0.0;
    if (gather) eb = fill((double)t128 / 512.0) * ((double)a.M / (((a.M + 127) / 128) * 128.0));     // the gathered form has the 128x128 kernel only
    else choose_gemm_kernel(a, &eb);
    int best = 0;
    for (int r1 = 3; r1 >= 2; --r1) {
        const int bm = 128 + 32 * r1;
        const long t7 = (long)((a.M + bm - 1) / bm) * (a.N / GEMM7_BN);
        if (t7 > 256) continue;                                          // one round only: beyond it the 256-row kernels' rate wins
        const double e7 = fill((double)t7 / 256.0) * ((double)a.M / (((a.M + bm - 1) / bm) * (double)bm)) * GEMM7_RATE;
        if (e7 > eb * 1.03) { best = r1; eb = e7; }
    }
    return best;
}
template <int ACT, bool F32, bool G, int R1, bool WEAVE = false>
static void launch_gemm7(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    constexpr int bm = 128 + 32 * R1, lds = Gemm7Geo<R1>::LDS_BYTES;
    a.tiles_m = (a.M + bm - 1) / bm;
    a.tiles_n = a.N / GEMM7_BN;
    lds_attr<gemm7_bf16_kernel<ACT, F32, G, R1, WEAVE>>(lds);
    hipLaunchKernelGGL((gemm7_bf16_kernel<ACT, F32, G, R1, WEAVE>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, s, a);
}

// Split-K factor for the 128x128 kernel (1 = do not split).  Only for grids that leave most resident slots empty: a lone
// workgroup streams its K-tiles at ~0.64 us each (0.95 us when two share a CU), so a 96-tile grid with K = 4096 takes 42 us
// however idle the chip is (scripts/kernel_bench.py --small).  Cost model in us per launch, d over the divisors of the
// K-tile count: (K-tiles / d) x per-tile time at the resulting occupancy + partial write/reduce.
static int choose_splitk(const GemmArgs& a, const GemmCtl& c) {
    if (!c.ws || !c.splitk || (c.variant != 0 && c.variant != 1)) return 1;
    const int tiles = a.tiles_m * a.tiles_n, nt = a.K / GEMM_BK;
    if (tiles > SPLITK_MAX_TILES || nt < 32) return 1;       // measured: K = 1024 GEMMs lose (14.0 -> 19.7 us)
    int best = 1;
    double cb = 1e30;
    for (int d = 1; d <= 32 && d * tiles <= SPLITK_MAX_WG; ++d) {
        if (nt % d || nt / d < 4) continue;
        const int wg = tiles * d;
        const double per = wg <= 256 ? 0.64 : 0.95 * ((wg + 511) / 512);
        const double cst = (nt / d) * per + (d > 1 ? 2.0 + 0.3 * d : 0.0);
        if (cst < cb * (d > 1 ? 0.85 : 1.0)) { cb = cst; best = d; }
    }
    return best;
}

// Small-M form (64x64 tiles, gemm_s_bf16_kernel): when the 128x128 grid cannot even give every CU one tile, quartering the
// tile spreads the operand stream over the idle CUs (a workgroup streams at ~55 GB/s whatever the chip does).  Same K order
// as every other kernel -> same bits.  Measured crossover (scripts/kernel_bench.py --small): see profiles/.
static bool want_small_m(const GemmArgs& a, const GemmCtl& c) {
    if (c.variant == 32) return true;
    if (c.variant != 0) return false;
    return a.tiles_m * a.tiles_n <= 128 && a.K >= 512;   // measured crossover: wins at <= 128 tiles, loses at 152-160
}

// Epilogue form of the two ping-pong kernels (k_gemm.h gemm_store_tr): bf16 outputs WITHOUT a residual go through the
// register-resident C^T epilogue (measured on MI355X, scripts/ubench/gemm_lab.hip: fc1 + QuickGELU 101.8 -> 95.6 us, gate/up
// 393 -> 384 us, STC 1x1 convs -1.5..-4 %), outputs with a residual keep the LDS-transposing one (the residual's row-contiguous
// 16-B loads, all requested up front, beat the 8-B pieces the C^T layout needs: 31.2 vs 35.8 us on the ViT out_proj).  Both forms
// produce the same bits (hash-checked per shape), so the choice is invisible to every caller.
static bool want_tr_epilogue(const GemmArgs& a) { return a.res == nullptr; }

#ifdef VL2_LAB
// gemm8 (k_gemm8.h): the 256 x 256 tile on four waves, one per SIMD, 128 x 128 wave tiles
template <int ACT, bool SW, bool F32>
static void launch_gemm8(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = a.N / GEMM4_BN;
    if constexpr (!F32) {
        if (want_tr_epilogue(a)) {
            lds_attr<gemm8_bf16_kernel<ACT, SW, false, true>>(GEMM8_LDS_BYTES);
            hipLaunchKernelGGL((gemm8_bf16_kernel<ACT, SW, false, true>), dim3(a.tiles_m * a.tiles_n), dim3(256), GEMM8_LDS_BYTES, s, a);
            return;
        }
    }
    lds_attr<gemm8_bf16_kernel<ACT, SW, F32, false>>(GEMM8_LDS_BYTES);
    hipLaunchKernelGGL((gemm8_bf16_kernel<ACT, SW, F32, false>), dim3(a.tiles_m * a.tiles_n), dim3(256), GEMM8_LDS_BYTES, s, a);
}
#endif

// gemm9 (k_gemm9.h): the 256 x 256 ping-pong tile on the 16 x 16 x 32 matrix instruction -- opt-in, NOT the family's bits
template <bool SW>
static void launch_gemm9(const GemmArgs& a0, int mode, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = a.N / GEMM4_BN;
    const dim3 grid(a.tiles_m * a.tiles_n);
    if (mode == 9) {          // variant 26: 64-deep phases, 5-stage ring = the whole 160 KiB of LDS
        lds_attr<gemm9_bf16_kernel<SW, 9>>(5 * GEMM4_STAGE);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 9>), grid, dim3(512), 5 * GEMM4_STAGE, s, a);
        return;
    }
    if constexpr (kLab) {
    if (mode == 1) {          // lab (variant 17): the LDS-DMA issue behind the load phase's fragment reads
        lds_attr<gemm9_bf16_kernel<SW, 1>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 1>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (mode == 2) {   // lab (variant 18): woven into the matrix phase
        lds_attr<gemm9_bf16_kernel<SW, 2>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 2>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (mode == 7) {   // lab (variant 23): variant 16 with s_memtime stamps, sums into the workspace (scripts/gemm9_phase_stamps.py)
        lds_attr<gemm9_bf16_kernel<SW, 7>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 7>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (mode == 8) {   // lab (variant 25): variant 16 with one stamp pair around the K loop
        lds_attr<gemm9_bf16_kernel<SW, 8>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 8>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (mode == 4) {   // lab (variant 20): two pieces at the head of the load phase, two woven into the matrix phase
        lds_attr<gemm9_bf16_kernel<SW, 4>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 4>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (mode == 5) {   // lab (variant 21): one / three
        lds_attr<gemm9_bf16_kernel<SW, 5>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 5>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (mode == 6) {   // lab (variant 22): three / one
        lds_attr<gemm9_bf16_kernel<SW, 6>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 6>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else if (mode == 3) {   // lab (variant 19): register-staged slabs (plain loads + ds_write_b128), no LDS-DMA
        lds_attr<gemm9_bf16_kernel<SW, 3>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 3>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    }
    if (mode != 0) return;
    }
    lds_attr<gemm9_bf16_kernel<SW, 0>>(GEMM4_LDS_BYTES);
    hipLaunchKernelGGL((gemm9_bf16_kernel<SW, 0>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
}

template <int ACT, bool SW, bool F32, int BM = GEMM4_BM>
static void launch_gemm4(const GemmArgs& a0, hipStream_t s, bool weave4 = false) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = a.N / GEMM4_BN;
    const dim3 grid(a.tiles_m * a.tiles_n);
    // the product builds ONE issue order per tile height: woven (WEAVE4) on the 192-row tiles, load-phase on the 256-row tiles; the other
    // order of each is a lab form (VL2_GEMM_NO_WEAVE4 / VL2_GEMM_WEAVE4)
    constexpr bool kWoven = BM == 192 && !F32, kPlain = !kWoven;
    if constexpr (!kLab) weave4 = kWoven;
    if constexpr (BM == 160) {                       // one form: LDS epilogue, load-phase issue
        lds_attr<gemm4_bf16_kernel<ACT, SW, F32, false, -1, 160>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, F32, false, -1, 160>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    } else {
    if constexpr (!F32) {
        if (want_tr_epilogue(a)) {
            if constexpr (kLab || kWoven) {
                if (weave4) {
                    lds_attr<gemm4_bf16_kernel<ACT, SW, false, true, -1, BM, true>>(GEMM4_LDS_BYTES);
                    hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false, true, -1, BM, true>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
                    return;
                }
            }
            if constexpr (kLab || kPlain) {
                lds_attr<gemm4_bf16_kernel<ACT, SW, false, true, -1, BM>>(GEMM4_LDS_BYTES);
                hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false, true, -1, BM>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
            }
            return;
        }
        if constexpr (kLab || kWoven) {
            if (weave4) {
                lds_attr<gemm4_bf16_kernel<ACT, SW, false, false, -1, BM, true>>(GEMM4_LDS_BYTES);
                hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, false, false, -1, BM, true>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
                return;
            }
        }
    }
    if constexpr (kLab || kPlain) {
        lds_attr<gemm4_bf16_kernel<ACT, SW, F32, false, -1, BM>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm4_bf16_kernel<ACT, SW, F32, false, -1, BM>), grid, dim3(512), GEMM4_LDS_BYTES, s, a);
    }
    }
}

// ---- persistent ping-pong GEMM (k_gemm6.h): ONE workgroup per CU walks its tiles, the LDS ring runs across tile boundaries, the stores
// of a tile are never waited for, the epilogue vectors arrive by LDS-DMA.  For bf16 outputs without residual / statistics / gather / remap
// whose 256-row grid is more than one round of workgroups (a one-round grid has nothing to overlap).  Measured in the C++ lab on one box,
// interleaved with the one-tile-per-workgroup kernels and bit-identical to them (profiles/r04_gemm_lab_persistent.txt): ViT q/k/v
// 9232x3072x1024 65.8 -> 59.6 us, ViT fc1 + QuickGELU 94.5 -> 87.3 (101.8 -> 89.9 with the LayerNorm carried), STC K = 1024 conv
// 81.8 -> 75.8, STC 4096x4096 conv 258.8 -> 254.0, 8192x4096x4096 216.2 -> 212.3 (1295 TF/s).
static int cu_count() {
    static std::atomic<int> n{0};                                  // idempotent cache of a device constant (like lds_attr's bits)
    int v = n.load(std::memory_order_relaxed);
    if (v == 0) {
        int dev = 0;
        check(hipGetDevice(&dev));
        check(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        if (v <= 0) v = 256;
        n.store(v, std::memory_order_relaxed);
    }
    return v;
}
static bool gemm6_ok(const GemmArgs& a, int bm) {
    return a.res == nullptr && a.stats_out == nullptr && a.a_idx == nullptr && a.out_grp == 0 && a.res_row_mod == 0 &&
           (a.norm == 0 || a.row_norm != nullptr) && a.N % GEMM4_BN == 0 && a.K % GEMM4_BK == 0 && a.K >= 1024 && a.M >= bm;
}
// 0 = not this call; 60 = 256-row tiles, 61 = 192-row tiles: the smaller makespan in units of a 256-row tile (a 192-row tile costs 0.75)
static int choose_gemm6(const GemmArgs& a) {
    // only with a counter block (dynamic tile hand-out): the static walk is as slow as its slowest CU (k_gemm6.h)
    if (!gemm6_ok(a, 256) || a.tile_ctr == nullptr) return 0;
    const int cus = cu_count();
    const long t256 = (long)((a.M + 255) / 256) * (a.N / GEMM4_BN), t192 = (long)((a.M + 191) / 192) * (a.N / GEMM4_BN);
    if (t256 <= cus) return 0;
    const double ms256 = (double)((t256 + cus - 1) / cus), ms192 = 0.75 * (double)((t192 + cus - 1) / cus);
    return ms192 < ms256 ? 71 : 70;
}
// kern: 60 / 61 / 62 static tile walk (256- / 192-row tiles / 192 rows + two accumulator sets), 70 / 71 = 60 / 61 through the counter block
template <int ACT, bool SW>
static void launch_gemm6(const GemmArgs& a0, int kern, hipStream_t s) {
    GemmArgs a = a0;
    a.tile_first_dyn = kern >= 80;                                     // 80 / 81: the first tile of a workgroup from the counter too
    if (kern >= 80) kern -= 20; else if (kern >= 70) kern -= 10; else a.tile_ctr = nullptr;
    const int bm = kern == 60 ? 256 : 192;
    a.tiles_m = (a.M + bm - 1) / bm;
    a.tiles_n = a.N / GEMM4_BN;
    const long nt = (long)a.tiles_m * a.tiles_n;
    const int g = (int)(nt < cu_count() ? nt : cu_count());
    if (kern == 60) {
        lds_attr<gemm6_bf16_kernel<ACT, SW, 256, false>>(GEMM6_LDS_BYTES);
        hipLaunchKernelGGL((gemm6_bf16_kernel<ACT, SW, 256, false>), dim3(g), dim3(512), GEMM6_LDS_BYTES, s, a);
    } else if (kern == 61) {
        lds_attr<gemm6_bf16_kernel<ACT, SW, 192, false>>(GEMM6_LDS_BYTES);
        hipLaunchKernelGGL((gemm6_bf16_kernel<ACT, SW, 192, false>), dim3(g), dim3(512), GEMM6_LDS_BYTES, s, a);
    } else if constexpr (kLab) {                                       // 62: two accumulator sets (measured slower than 61)
        lds_attr<gemm6_bf16_kernel<ACT, SW, 192, true>>(GEMM6_LDS_BYTES);
        hipLaunchKernelGGL((gemm6_bf16_kernel<ACT, SW, 192, true>), dim3(g), dim3(512), GEMM6_LDS_BYTES, s, a);
    }
}

// Row split (returns the number of leading rows that go to the 256x256 kernel, 0 = no split).  Two cases, both two launches on the
// same stream; every kernel accumulates K in the same order and shares one epilogue, so the output bits do not change (asserted
// in tests/test_gpu_ops.py):
//  (1) M = 1621 leaves the 256-row kernel a 7th row tile with 85 live rows (9.5 % of its MFMA work wasted) and the 128-row kernel
//      is slower per FLOP.  When N is wide enough for both parts to fill the chip, the first floor(M/256)*256 rows go to the
//      256x256 kernel and the remaining rows to whatever the chooser picks for them.  Measured: 1621x28672x4096 + SwiGLU
//      362 -> 345 us; loses on N <= 6144 and on M % 256 > 128.
//  (2) whole rounds + a short tail: 9216x4096x4096 (the STC s1 convolutions) is 576 tiles = 2.25 rounds of 256 workgroups; the
//      quarter round costs most of a full one.  The rows of the whole rounds stay on the 256x256 kernel, the tail rows (at most
//      half a round of big tiles) go through the chooser, which gives them a finer-tiled one-round kernel
//      (2 rounds 224 us + 1024 rows on the 8-wave 128x128 kernel ~40 us against 287 us for the single launch).
static int m_split_rows(const GemmArgs& a, const GemmCtl& c) {
    if (c.variant != 0 || a.out_grp > 0 || a.res_row_mod > 0 || a.N % GEMM4_BN || a.K < 1024 || a.M < 1024) return 0;
    const int r = a.M % GEMM4_BM, m1_tiles = a.M / GEMM4_BM, n_tiles = a.N / GEMM4_BN;
    if (r != 0 && r <= 96 && n_tiles >= 64) {
        const long t4 = (long)m1_tiles * n_tiles;
        if ((double)t4 / (double)(((t4 + 255) / 256) * 256) >= 0.85) return m1_tiles * GEMM4_BM;   // the 256-row part fills its rounds
    }
    if (256 % n_tiles == 0) {
        const int per_round = 256 / n_tiles;                       // row tiles in one full round of 256 workgroups
        const int mt = (a.M + GEMM4_BM - 1) / GEMM4_BM, rounds = mt / per_round, tail = mt - rounds * per_round;
        if (rounds >= 1 && tail > 0 && tail * n_tiles <= 128) return rounds * per_round * GEMM4_BM;
    }
    return 0;
}

// rows [m0, m0 + rows) of a call as a call of its own (A / C / residual / statistics rows shifted; the gather table keeps its
// row stride idx_ld)
static GemmArgs gemm_rows(const GemmArgs& a0, int m0, int rows, bool f32) {
    GemmArgs a = a0;
    a.M = rows;
    if (a0.a_idx) a.a_idx = a0.a_idx + m0; else a.A = a0.A + (size_t)m0 * a0.lda;
    a.C = f32 ? (void*)((float*)a0.C + (size_t)m0 * a0.ldc) : (void*)((bf16_t*)a0.C + (size_t)m0 * a0.ldc);
    if (a0.res) a.res = a0.res + (size_t)m0 * a0.ldres;
    if (a0.stats_out) a.stats_out = a0.stats_out + (size_t)m0 * a0.stats_out_np * 2;
    if (a0.stats_in) a.stats_in = a0.stats_in + (size_t)m0 * a0.stats_in_np * 2;
    if (a0.row_norm) a.row_norm = a0.row_norm + (size_t)m0 * 2;
    if (a0.row_norm_out) { a.row_norm_out = a0.row_norm_out + (size_t)m0 * 2; a.row_ticket = a0.row_ticket + m0 / 64; }   // 64 = the smallest tile height
    a.tiles_m = (rows + GEMM_BM - 1) / GEMM_BM;
    return a;
}

template <int ACT, bool SW, bool F32, bool G>
static void launch_gemm(const GemmArgs& a0, const GemmCtl& c, hipStream_t s) {
    if constexpr (!G && !F32 && ACT == ACT_NONE) {
        if (c.mfma16 && (c.variant == 23 || c.variant == 25)) {                         // the stamps need room for 8 waves x 6 u64 per workgroup
            GemmArgs a = a0;
            const long wgs = (long)((a.M + 255) / 256) * (a.N / 256);
            a.sk_ws = (c.ws && c.ws_bytes >= wgs * 8 * 6 * 8) ? (float*)c.ws : nullptr;
            launch_gemm9<SW>(a, c.variant == 23 ? 7 : 8, s);
            return;
        }
        if (c.mfma16) {                                                                                                                         // (vl2_gemm has checked that the call qualifies)
            // a row-split call keeps its ONE mixed launch (k_gemm9.h gemm_mix16_bf16_kernel): whole 256-row tiles on gemm9_body, the tail rows on the
            // 128 x 128 body of the same instruction -- the same bits row by row, so the split stays invisible (as in the 32 x 32 x 16 family)
            if (c.variant == 0 || c.variant == 16 || c.variant == 26 || (kLab && c.variant >= 27 && c.variant <= 29)) {
                GemmCtl c0 = c;
                c0.variant = 0;
                if (const int M1 = m_split_rows(a0, c0); M1 > 0 && !c.no_mix) {
                    GemmArgs big = gemm_rows(a0, 0, M1, false), tail = gemm_rows(a0, M1, a0.M - M1, false);
                    if (kLab && c.variant >= 27) big.tile_group = c.variant == 27 ? 8 : c.variant == 28 ? 0 : 6;      // lab: variant 26 with another group depth
                    if (kLab && c.variant == 28) tail.tile_group = -1;                                                // lab: variant 26 with the tail's empty waves computing (as before round 6's skip)
                    const long t_big = (long)(M1 / GEMM4_BM) * (a0.N / GEMM4_BN), t_tail = (long)tail.tiles_m * tail.tiles_n;
                    if (t_tail > 128 && t_tail <= 512) {
                        big.tiles_m = M1 / GEMM4_BM; big.tiles_n = a0.N / GEMM4_BN;
                        const dim3 gmix((unsigned)(t_big + t_tail));
                        // the big tiles with 64-deep phases on the 5-stage ring (the whole 160 KiB of LDS): the flag's form and variant 26 (gate/up at
                        // S = 945 / 1621 / 2973: 195.9 / 334.8 / 609.5 -> 192.4 / 332.2 / 596.9 us, profiles/r06_mix16_bench_64deep.txt); 16 = the 32-deep form
                        // (SwiGLU calls only: without SwiGLU the tail body ends in gemm_rows_ticket, whose LDS word does not fit beside the 160-KiB ring)
                        if constexpr (SW) {
                            if (c.variant != 16) {
                                lds_attr<gemm_mix16_bf16_kernel<SW, 9>>(5 * GEMM4_STAGE);
                                hipLaunchKernelGGL((gemm_mix16_bf16_kernel<SW, 9>), gmix, dim3(512), 5 * GEMM4_STAGE, s, big, tail, (int)t_big);
                                return;
                            }
                        }
                        lds_attr<gemm_mix16_bf16_kernel<SW>>(GEMM4_LDS_BYTES);
                        hipLaunchKernelGGL((gemm_mix16_bf16_kernel<SW>), gmix, dim3(512), GEMM4_LDS_BYTES, s, big, tail, (int)t_big);
                        return;
                    }
                }
            }
            if constexpr (!SW) {
                // the other two tiles of the 16 x 16 x 32 set (round 6: the decoder's o / down projections): at most one 128 x 128 tile per CU -> the one-round
                // 128 x 128 body; a one-round grid of fill-the-round tiles where the family's rule picks them -> gemm7 on this instruction; else the 256 x 256 tile
                if (c.variant == 0 || c.variant == 256 || c.variant == 224 || c.variant == 192) {       // (256 / 224 / 192 with the flag: that tile of the set on demand)
                    if (c.variant == 256 || (c.variant == 0 && (long)a0.tiles_m * a0.tiles_n <= 256)) {
                        lds_attr<gemm_l8_16_bf16_kernel<false>>(GEMML_LDS_BYTES);
                        hipLaunchKernelGGL((gemm_l8_16_bf16_kernel<false>), dim3(a0.tiles_m * a0.tiles_n), dim3(512), GEMML_LDS_BYTES, s, a0);
                        return;
                    }
                    // (lab: the fill-the-round tiles on this instruction, k_gemm7.h gemm7_loop16 -- bit-identical with the rest of the set, measured at parity
                    //  with their 32 x 32 x 16 twins on the one-round grids they exist for: no call site uses them)
                    if (const int r1 = !kLab ? 0 : c.variant == 224 ? 3 : c.variant == 192 ? 2 : c.no_fill ? 0 : choose_gemm7(a0, false); r1 != 0) {
                        GemmArgs a = a0;
                        const int bm = 128 + 32 * r1;
                        a.tiles_m = (a.M + bm - 1) / bm;
                        a.tiles_n = a.N / GEMM7_BN;
#ifdef VL2_LAB
                        if (r1 == 3) {
                            lds_attr<gemm7_16_bf16_kernel<3>>(Gemm7Geo<3>::LDS_BYTES);
                            hipLaunchKernelGGL((gemm7_16_bf16_kernel<3>), dim3(a.tiles_m * a.tiles_n), dim3(512), Gemm7Geo<3>::LDS_BYTES, s, a);
                        } else {
                            lds_attr<gemm7_16_bf16_kernel<2>>(Gemm7Geo<2>::LDS_BYTES);
                            hipLaunchKernelGGL((gemm7_16_bf16_kernel<2>), dim3(a.tiles_m * a.tiles_n), dim3(512), Gemm7Geo<2>::LDS_BYTES, s, a);
                        }
                        return;
#endif
                    }
                }
            }
            if (c.variant != 0 && c.variant != 16 && a0.row_norm_out && c.fin) *c.fin = true;      // only the shipped form (MODE 0) finalizes its rows itself
            launch_gemm9<SW>(a0, c.variant >= 17 && c.variant <= 23 ? c.variant - 16 : c.variant == 26 ? 9 : 0, s);
            return;
        }
    }
    if constexpr (!SW) {
        // fill-the-round tiles (k_gemm7.h): variants 224 / 192 on request (any shape with N % 128 == 0), or by the rule of choose_gemm7
        const int r1 = c.variant == 224 ? 3 : c.variant == 192 ? 2 : (c.variant == 0 && !c.no_fill) ? choose_gemm7(a0, G) : 0;
        // (225 / 193 = 224 / 192 with the LDS-DMA issue woven into the MFMA phases, as VL2_GEMM_WEAVE selects it: lab form, see gemm3 below)
        const int r1v = c.variant == 225 ? 3 : c.variant == 193 ? 2 : r1;
        const bool weave7 = kLab && (c.weave || c.variant == 225 || c.variant == 193);
        if constexpr (kLab) {
            if (weave7 && r1v == 3) { launch_gemm7<ACT, F32, G, 3, true>(a0, s); return; }
            if (weave7 && r1v == 2) { launch_gemm7<ACT, F32, G, 2, true>(a0, s); return; }
        }
        if (r1v == 3) { launch_gemm7<ACT, F32, G, 3, false>(a0, s); return; }
        if (r1v == 2) { launch_gemm7<ACT, F32, G, 2, false>(a0, s); return; }
    }
    if constexpr (!G && !F32) {
        // persistent form: on request (variants 60 / 61; 62 = 192-row tiles with two accumulator sets, measured slower, kept for the lab) or
        // by the rule of choose_gemm6; a forced variant the call does not qualify for falls through to the automatic choice below
        const int v6 = c.variant;
        const int k6 = v6 == 0 ? (c.persist && !c.no_persist ? choose_gemm6(a0) : 0)
                     : (v6 == 60 || v6 == 61 || v6 == 62 || ((v6 == 70 || v6 == 71 || v6 == 80 || v6 == 81) && a0.tile_ctr)) &&
                               gemm6_ok(a0, v6 == 60 || v6 == 70 || v6 == 80 ? 256 : 192) &&
                               (v6 != 62 || a0.K >= 32 * 28) ? v6 : 0;
        if (k6) { launch_gemm6<ACT, SW>(a0, k6, s); return; }
        if (v6 == 24 || (v6 >= 60 && v6 <= 81)) {   // 24 = the automatic choice WITHOUT the persistent form (A/B)
            GemmCtl c0 = c;
            c0.variant = 0;
            c0.no_persist = true;
            launch_gemm<ACT, SW, F32, G>(a0, c0, s);
            return;
        }
    }
    if constexpr (!G && !F32) {
        // (K < 2048 with an activation in the epilogue -- ViT fc1 + QuickGELU -- stays on its single launch: in the pipeline's rocprofv3 trace the
        //  mixed form took 105 us against 98.6 for the 128x128 kernel, although it wins the isolated micro-benchmark; end to end the two are equal:
        //  encode 12.52 vs 12.52 ms over three alternations)
        if (const int M1 = (a0.K < 2048 && ACT != ACT_NONE) ? 0 : m_split_rows(a0, c); M1 > 0) {
            GemmArgs big = gemm_rows(a0, 0, M1, F32), tail = gemm_rows(a0, M1, a0.M - M1, F32);
            // ONE launch (k_gemm.h gemm_mix_bf16_kernel) when the tail is 129..512 tiles of the 8-wave 128x128 body: its workgroups start on
            // the CUs that run out of big tiles instead of behind a second launch.  Measured against the two launches on one box
            // (scripts/gpu_r3_k.sh): gate/up at S = 1621 383.0 -> 369.7 us (prefill 26.0 -> 25.6 ms), STC 4096x4096 conv 274.7 -> 265.0,
            // STC K = 1024 conv 85.9 -> 82.2, ViT fc1 99.2 -> 96.5 (T = 8: 54.7 -> 49.9); in the pipeline the K = 1024 cases are neutral at T = 16
            // (encode 12.90 vs 12.89 ms) and the prefill gains 0.4 ms.  VL2_GEMM_NO_MIX in the descriptor keeps two launches / the single kernel.
            const bool no_mix = c.no_mix;
            const long t_big = (long)(M1 / GEMM4_BM) * (a0.N / GEMM4_BN), t_tail = (long)tail.tiles_m * tail.tiles_n;
            if (!no_mix && t_tail > 128 && t_tail <= 512) {
                big.tiles_m = M1 / GEMM4_BM; big.tiles_n = a0.N / GEMM4_BN;
                const dim3 gmix((unsigned)(t_big + t_tail));
                if constexpr (kLab) {                 // the big tiles with the woven LDS-DMA issue (VL2_GEMM_WEAVE4)
                    if (c.weave4) {
                        if (want_tr_epilogue(big)) {
                            lds_attr<gemm_mix_bf16_kernel<ACT, SW, true, true>>(GEMM4_LDS_BYTES);
                            hipLaunchKernelGGL((gemm_mix_bf16_kernel<ACT, SW, true, true>), gmix, dim3(512), GEMM4_LDS_BYTES, s, big, tail, (int)t_big);
                        } else {
                            lds_attr<gemm_mix_bf16_kernel<ACT, SW, false, true>>(GEMM4_LDS_BYTES);
                            hipLaunchKernelGGL((gemm_mix_bf16_kernel<ACT, SW, false, true>), gmix, dim3(512), GEMM4_LDS_BYTES, s, big, tail, (int)t_big);
                        }
                        return;
                    }
                }
                if (want_tr_epilogue(big)) {
                    lds_attr<gemm_mix_bf16_kernel<ACT, SW, true>>(GEMM4_LDS_BYTES);
                    hipLaunchKernelGGL((gemm_mix_bf16_kernel<ACT, SW, true>), gmix, dim3(512), GEMM4_LDS_BYTES, s, big, tail, (int)t_big);
                } else {
                    lds_attr<gemm_mix_bf16_kernel<ACT, SW, false>>(GEMM4_LDS_BYTES);
                    hipLaunchKernelGGL((gemm_mix_bf16_kernel<ACT, SW, false>), gmix, dim3(512), GEMM4_LDS_BYTES, s, big, tail, (int)t_big);
                }
                return;
            }
            if (a0.K >= 2048) {                       // two launches (measured wins at K >= 2048 only: profiles/r02_experiments.md section 3)
                launch_gemm4<ACT, SW, F32>(big, s, c.weave4);
                launch_gemm<ACT, SW, F32, G>(tail, c, s);
                return;
            }
        }
    }
    if constexpr (!SW && !F32) {
        if (choose_splitk(a0, c) <= 1 && want_small_m(a0, c)) {
            if (a0.row_norm_out && c.fin) *c.fin = true;
            lds_attr<gemm_s_bf16_kernel<ACT, G>>(GEMMS_LDS_BYTES);
            GemmArgs a = a0;
            a.tiles_m = (a.M + GEMMS_BM - 1) / GEMMS_BM;
            a.tiles_n = a.N / GEMMS_BN;
            hipLaunchKernelGGL((gemm_s_bf16_kernel<ACT, G>), dim3(a.tiles_m * a.tiles_n), dim3(128), GEMMS_LDS_BYTES, s, a);
            return;
        }
    }
    if (const int split = choose_splitk(a0, c); split > 1) {
        if (a0.row_norm_out && c.fin) *c.fin = true;
        lds_attr<gemm_bf16_kernel<ACT, SW, F32, G, false, true>>(GEMM_LDS_BYTES);
        GemmArgs a = a0;
        a.sk_ws = (float*)c.ws;
        a.sk_flags = (int*)((char*)c.ws + SPLITK_CNT_OFF);
        hipLaunchKernelGGL((gemm_bf16_kernel<ACT, SW, F32, G, false, true>), dim3(a.tiles_m * a.tiles_n, split), dim3(256), GEMM_LDS_BYTES, s, a);
        return;
    }
    if constexpr (!G) {
        // measured (scripts/kernel_bench.py --frames 8): wins 15-20 % at <= 256 tiles with K >= 4096, loses at K = 1024 and beyond one round
        if (c.variant == 256 || (c.variant == 0 && (long)a0.tiles_m * a0.tiles_n <= 256 && a0.K >= 4096)) {
            lds_attr<gemm_l8_bf16_kernel<ACT, SW, F32>>(GEMML_LDS_BYTES);
            hipLaunchKernelGGL((gemm_l8_bf16_kernel<ACT, SW, F32>), dim3(a0.tiles_m * a0.tiles_n), dim3(512), GEMML_LDS_BYTES, s, a0);
            return;
        }
        const int kern = c.variant == 0 ? choose_gemm_kernel(a0) : c.variant;
        if ((kern == 4 || kern == 5) && a0.N % GEMM3_BN == 0) {
            // WEAVE (k_gemm.h gemm3_body; same bits): the LDS-DMA issue woven into the MFMA phases.  Measured round 5 (profiles/r05_experiments.md):
            // -2...-9 % back to back with the operands warm in the Infinity Cache, but +4...+12 % IN THE PIPELINE (down 206 -> 218 us, o 66 -> 69,
            // Conv3d on the 192-row tiles 425 -> 476): the woven pieces have 1.5 phases of flight instead of 3, which cold weights do not forgive.
            // So it is a lab switch: variant 5 or VL2_GEMM_WEAVE.
            const bool weave = kLab && (kern == 5 || c.weave);
            GemmArgs a = a0;
            a.tiles_m = (a.M + GEMM3_BM - 1) / GEMM3_BM;
            a.tiles_n = a.N / GEMM3_BN;
            const dim3 grid(a.tiles_m * a.tiles_n);
            if constexpr (kLab) {
                if (weave) {
                    if constexpr (!F32) {
                        if (want_tr_epilogue(a)) {
                            lds_attr<gemm3_bf16_kernel<ACT, SW, false, true, -1, true>>(GEMM3_LDS_BYTES);
                            hipLaunchKernelGGL((gemm3_bf16_kernel<ACT, SW, false, true, -1, true>), grid, dim3(512), GEMM3_LDS_BYTES, s, a);
                            return;
                        }
                    }
                    lds_attr<gemm3_bf16_kernel<ACT, SW, F32, false, -1, true>>(GEMM3_LDS_BYTES);
                    hipLaunchKernelGGL((gemm3_bf16_kernel<ACT, SW, F32, false, -1, true>), grid, dim3(512), GEMM3_LDS_BYTES, s, a);
                    return;
                }
            }
            if constexpr (!F32) {
                if (want_tr_epilogue(a)) {
                    lds_attr<gemm3_bf16_kernel<ACT, SW, false, true>>(GEMM3_LDS_BYTES);
                    hipLaunchKernelGGL((gemm3_bf16_kernel<ACT, SW, false, true>), grid, dim3(512), GEMM3_LDS_BYTES, s, a);
                    return;
                }
            }
            lds_attr<gemm3_bf16_kernel<ACT, SW, F32>>(GEMM3_LDS_BYTES);
            hipLaunchKernelGGL((gemm3_bf16_kernel<ACT, SW, F32>), grid, dim3(512), GEMM3_LDS_BYTES, s, a);
            return;
        }
#ifdef VL2_LAB
        if (kern == 9 && a0.N % GEMM4_BN == 0) {                              // lab / forced: the four-wave 256 x 256 kernel (k_gemm8.h)
            launch_gemm8<ACT, SW, F32>(a0, s);
            return;
        }
#endif
        if ((kern == 8 || (F32 && kern == 12)) && a0.N % GEMM4_BN == 0) {     // (the 192-row form is built for bf16 outputs only)
            launch_gemm4<ACT, SW, F32>(a0, s, c.weave4);
            return;
        }
        if constexpr (!F32 && !SW && kLab) {
            if (kern == 10 && a0.N % GEMM4_BN == 0) {                          // lab: 160-row tiles (measured: not faster than the 192-row tiles)
                launch_gemm4<ACT, SW, false, 160>(a0, s);
                return;
            }
        }
        if constexpr (!F32) {
            if (kern == 12 && a0.N % GEMM4_BN == 0) {
                // 192-row tiles take the woven LDS-DMA issue by default (round 5, scripts/weave4_bench.py, interleaved on one box, same bits): their load
                // phase carries 3-4 pieces + 10 fragment reads under a partner's TWELVE MFMAs -- ViT fc2 82.5 -> 79.4 us, out_proj 30.1 -> 29.3, the
                // decoder's q/k/v 82.0 -> 81.6; the 256-row bodies measure the same either way (16 MFMAs cover the load phase) and keep the load-phase issue
                launch_gemm4<ACT, SW, false, 192>(a0, s, !c.no_weave4);
                return;
            }
        }
    }
    if constexpr (!G && kLab) {
        if (want_stream_k(a0, c)) {
            lds_attr<gemm_sk_bf16_kernel<ACT, SW, F32>>(GEMM_LDS_BYTES);
            GemmArgs a = a0;
            const int total = a.tiles_m * a.tiles_n * (a.K / GEMM_BK);
            a.sk_ws = (float*)c.ws;
            a.sk_flags = (int*)((char*)c.ws + SK_FLAGS_OFF);
            a.sk_per = (total + SK_GRID - 1) / SK_GRID;
            check(hipMemsetAsync(a.sk_flags, 0, (SK_GRID + 1) * 4, s));               // flags re-armed before EVERY launch (guide G16)
            hipLaunchKernelGGL((gemm_sk_bf16_kernel<ACT, SW, F32>), dim3(SK_GRID), dim3(256), GEMM_LDS_BYTES, s, a);
            return;
        }
    }
    lds_attr<gemm_bf16_kernel<ACT, SW, F32, G>>(GEMM_LDS_BYTES);   // 64 KiB dynamic LDS needs the opt-in once per kernel instance
    hipLaunchKernelGGL((gemm_bf16_kernel<ACT, SW, F32, G>), dim3(a0.tiles_m * a0.tiles_n), dim3(256), GEMM_LDS_BYTES, s, a0);
}

// ---- fp8 form (VL2_GEMM_FP8): A and W are e4m3fn bytes; the kernels see a row of K bytes as K / 2 16-bit "elements" (k_gemm.h gemm3 / gemm4 FP8),
// so `a` arrives with K, lda, ldw already halved.  128 x 256, 256 x 256 or 192 x 256 ping-pong tiles by the efficiency model of the 16-bit choice
// (the 128 x 128 kernels are not built for fp8), LDS epilogue with the row table and the column scales.
template <int ACT, bool SW, bool F32>
static int32_t launch_gemm_fp8(const GemmArgs& a0, int variant, hipStream_t s) {
    GemmArgs a = a0;
    int kern = variant ? variant : choose_gemm_kernel(a);
    if (kern != 4 && kern != 8 && kern != 12) kern = 4;
    if (F32 && kern == 12) kern = 8;
    a.tiles_n = a.N / 256;
    if (kern == 4) {
        a.tiles_m = (a.M + GEMM3_BM - 1) / GEMM3_BM;
        lds_attr<gemm3_fp8_kernel<ACT, SW, F32>>(GEMM3_LDS_BYTES);
        hipLaunchKernelGGL((gemm3_fp8_kernel<ACT, SW, F32>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM3_LDS_BYTES, s, a);
    } else if (kern == 8) {
        a.tiles_m = (a.M + 255) / 256;
        lds_attr<gemm4_fp8_kernel<ACT, SW, F32, 256>>(GEMM4_LDS_BYTES);
        hipLaunchKernelGGL((gemm4_fp8_kernel<ACT, SW, F32, 256>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a);
    } else {
        if constexpr (!F32) {
            a.tiles_m = (a.M + 191) / 192;
            lds_attr<gemm4_fp8_kernel<ACT, SW, false, 192>>(GEMM4_LDS_BYTES);
            hipLaunchKernelGGL((gemm4_fp8_kernel<ACT, SW, false, 192>), dim3(a.tiles_m * a.tiles_n), dim3(512), GEMM4_LDS_BYTES, s, a);
        }
    }
    return 0;
}

// one chunk (every operand within the kernels' 32-bit buffer offsets) -> the right instantiation
static int32_t gemm_dispatch(const GemmArgs& a, const GemmCtl& c, int act, bool sw, bool f32, hipStream_t s) {
    const bool g = a.a_idx != nullptr;
    if (a.out_grp > 0 || a.res_row_mod > 0) {                      // row-remap epilogue (patch-embed): dedicated instantiation
        lds_attr<gemm_bf16_kernel<ACT_NONE, false, false, false, true>>(GEMM_LDS_BYTES);
        hipLaunchKernelGGL((gemm_bf16_kernel<ACT_NONE, false, false, false, true>), dim3(a.tiles_m * a.tiles_n), dim3(256), GEMM_LDS_BYTES, s, a);
        return 0;
    }
    if (sw) launch_gemm<ACT_NONE, true, false, false>(a, c, s);
    else if (g) {
        if (act == VL2_ACT_SILU) launch_gemm<ACT_SILU, false, false, true>(a, c, s);
        else launch_gemm<ACT_NONE, false, false, true>(a, c, s);
    } else if (f32) launch_gemm<ACT_NONE, false, true, false>(a, c, s);
    else {
        switch (act) {
            case VL2_ACT_NONE: launch_gemm<ACT_NONE, false, false, false>(a, c, s); break;
            case VL2_ACT_QGELU: launch_gemm<ACT_QGELU, false, false, false>(a, c, s); break;
            case VL2_ACT_GELU: launch_gemm<ACT_GELU, false, false, false>(a, c, s); break;
            case VL2_ACT_SILU: launch_gemm<ACT_SILU, false, false, false>(a, c, s); break;
            case VL2_ACT_GELU_TANH: launch_gemm<ACT_GELU_TANH, false, false, false>(a, c, s); break;
            default: return fail(VL2_E_UNSUPP, "vl2_gemm: unknown act %d", act);
        }
    }
    return 0;
}

extern "C" int32_t vl2_gemm(const vl2_gemm_desc* d, void* stream) {
    if (!d || d->size != sizeof(vl2_gemm_desc)) return fail(VL2_E_BADARG, "vl2_gemm: descriptor missing or of another ABI (size %u, expected %zu)", d ? d->size : 0u, sizeof(vl2_gemm_desc));
    const int M = d->M, N = d->N, K = d->K, act = d->act;
    if (!d->A || !d->W || !d->C || M <= 0 || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_gemm: null pointer or empty shape");
    if (N % 128 || K % 64) return fail(VL2_E_SHAPE, "vl2_gemm: need N%%128==0 and K%%64==0 (N=%d K=%d)", N, K);
    if ((d->lda % 8) || (d->ldw % 8) || (d->ldc % 8) || (d->res && (d->ldres % 8)) || !ALIGNED16(d->A) || !ALIGNED16(d->W) || !ALIGNED16(d->C) ||
        (d->res && !ALIGNED16(d->res)) || (d->bias && !ALIGNED16(d->bias)) || (d->w_colsum && !ALIGNED16(d->w_colsum)) ||
        (d->stats_in && !ALIGNED16(d->stats_in)))
        return fail(VL2_E_SHAPE, "vl2_gemm: pointers / leading dims must be 16-byte aligned");
    const bool sw = d->flags & VL2_GEMM_SWIGLU, f32 = d->flags & VL2_GEMM_OUT_F32, g = d->a_idx != nullptr;
    if (d->flags & VL2_GEMM_FP8) {
        // W8A8 on the fp8 matrix pipe: K counts e4m3fn elements = bytes; row scales through `row_norm` (vl2_quant_act_fp8), column scales `col_scale`
        if (g || d->out_grp > 0 || d->res_row_mod > 0 || d->stats_out || d->stats_in || d->w_colsum || (d->flags & (VL2_GEMM_SPLITK | VL2_GEMM_PERSISTENT)))
            return fail(VL2_E_UNSUPP, "vl2_gemm: the fp8 form supports plain rows with bias / activation / SwiGLU / residual only");
        if (!d->row_norm || !d->col_scale) return fail(VL2_E_BADARG, "vl2_gemm: the fp8 form needs row_norm (vl2_quant_act_fp8) and col_scale (vl2_pack_quant_fp8)");
        if (N % 256 || K % 128 || (d->lda % 16) || (d->ldw % 16)) return fail(VL2_E_SHAPE, "vl2_gemm: the fp8 form needs N%%256==0, K%%128==0, 16-byte rows (N=%d K=%d)", N, K);
        if (sw && (f32 || act != VL2_ACT_NONE || d->bias)) return fail(VL2_E_UNSUPP, "vl2_gemm: SWIGLU excludes bias/act/f32");
        if (f32 && act != VL2_ACT_NONE) return fail(VL2_E_UNSUPP, "vl2_gemm: f32 output supports act none");
        if (!ALIGNED16(d->col_scale) || (((uintptr_t)d->row_norm) & 7)) return fail(VL2_E_SHAPE, "vl2_gemm: col_scale must be 16-byte, row_norm 8-byte aligned");
        if ((int64_t)(M - 1) * d->lda + K >= ((int64_t)1 << 31) - 65536 || (int64_t)(N - 1) * d->ldw + K >= ((int64_t)1 << 31) - 65536)
            return fail(VL2_E_UNSUPP, "vl2_gemm: fp8 operands of 2 GiB or more are not chunked");
        const int v8 = d->variant;
        if (!(v8 == 0 || v8 == 4 || v8 == 8 || v8 == 12)) return fail(VL2_E_BADARG, "vl2_gemm: the fp8 form has variants 0 / 4 / 8 / 12 (got %d)", v8);
        GemmArgs a{};
        a.A = (const bf16_t*)d->A; a.W = (const bf16_t*)d->W; a.C = d->C; a.bias = d->bias; a.res = (const bf16_t*)d->res;
        a.M = M; a.N = N; a.K = K / 2; a.lda = d->lda / 2; a.ldw = d->ldw / 2; a.ldc = d->ldc; a.ldres = d->ldres;
        a.norm = VL2_NORM_RMS; a.row_norm = d->row_norm; a.col_scale = d->col_scale;
        a.stats_out_np = N / 64; a.stats_in_np = a.K / 64; a.idx_ld = M;
        hipStream_t s8 = ST(stream);
        if (sw) launch_gemm_fp8<ACT_NONE, true, false>(a, v8, s8);
        else if (f32) launch_gemm_fp8<ACT_NONE, false, true>(a, v8, s8);
        else if (act == VL2_ACT_NONE) launch_gemm_fp8<ACT_NONE, false, false>(a, v8, s8);
        else if (act == VL2_ACT_SILU) launch_gemm_fp8<ACT_SILU, false, false>(a, v8, s8);
        else return fail(VL2_E_UNSUPP, "vl2_gemm: the fp8 form supports act none / silu");
        return launched("vl2_gemm (fp8)");
    }
    if (d->col_scale) return fail(VL2_E_BADARG, "vl2_gemm: col_scale without VL2_GEMM_FP8");
    if (g && (d->seg_k <= 0 || d->seg_k % 64 || K % d->seg_k)) return fail(VL2_E_SHAPE, "vl2_gemm: bad gather segments");
    const bool remap = d->out_grp > 0 || d->res_row_mod > 0;
    if (remap && (sw || g || f32 || act != VL2_ACT_NONE || d->norm || d->stats_out)) return fail(VL2_E_UNSUPP, "vl2_gemm: row remap supports plain bf16 output only");
    if (sw && (f32 || g || act != VL2_ACT_NONE || d->bias)) return fail(VL2_E_UNSUPP, "vl2_gemm: SWIGLU excludes bias/act/f32/gather");
    if (g && f32) return fail(VL2_E_UNSUPP, "vl2_gemm: gather + f32 not built");
    if (g && act != VL2_ACT_NONE && act != VL2_ACT_SILU) return fail(VL2_E_UNSUPP, "vl2_gemm: gather supports act none/silu");
    if (f32 && act != VL2_ACT_NONE) return fail(VL2_E_UNSUPP, "vl2_gemm: f32 output supports act none");
    if (d->norm != VL2_NORM_NONE) {
        if (d->norm != VL2_NORM_RMS && d->norm != VL2_NORM_LN) return fail(VL2_E_BADARG, "vl2_gemm: unknown norm %d", d->norm);
        if ((!d->stats_in && !d->row_norm) || g || remap) return fail(VL2_E_BADARG, "vl2_gemm: a fused norm needs stats_in (or row_norm) and plain A rows");
        if (d->norm == VL2_NORM_LN && (!d->w_colsum || sw)) return fail(VL2_E_BADARG, "vl2_gemm: fused LayerNorm needs w_colsum (and excludes SWIGLU)");
        if (!d->row_norm && (K % 128)) return fail(VL2_E_SHAPE, "vl2_gemm: reducing stats_in in the GEMM needs K %% 128 == 0 (16-byte pairs of partials; K=%d): pass row_norm (vl2_row_norm_finalize)", K);
    }
    if (d->stats_out && (sw || f32)) return fail(VL2_E_UNSUPP, "vl2_gemm: stats_out needs a plain bf16 output");
    if (d->row_norm_out) {
        if (!d->stats_out || !d->row_ticket) return fail(VL2_E_BADARG, "vl2_gemm: row_norm_out needs stats_out and row_ticket");
        if (d->norm_out != VL2_NORM_RMS && d->norm_out != VL2_NORM_LN) return fail(VL2_E_BADARG, "vl2_gemm: unknown norm_out %d", d->norm_out);
        if ((((uintptr_t)d->row_norm_out) & 7) || (((uintptr_t)d->row_ticket) & 3)) return fail(VL2_E_SHAPE, "vl2_gemm: row_norm_out must be 8-byte, row_ticket 4-byte aligned");
    }
    if (d->ws && (d->ws_bytes < SK_WS_BYTES || !ALIGNED16(d->ws))) return fail(VL2_E_BADARG, "vl2_gemm: workspace needs >= %lld bytes, 16-byte aligned", (long long)SK_WS_BYTES);
    const int v = d->variant;
    if (!(v == 0 || v == 1 || v == 2 || v == 4 || v == 5 || v == 8 || v == 9 || v == 10 || v == 12 || (v >= 16 && v <= 23) || v == 25 || (v >= 26 && v <= 29) || v == 32 || v == 24 || v == 60 || v == 61 || v == 62 || v == 70 || v == 71 || v == 80 || v == 81 || v == 192 || v == 193 || v == 224 || v == 225 || v == 256)) return fail(VL2_E_BADARG, "vl2_gemm: unknown variant %d", v);
    if (!kLab && (v == 2 || v == 5 || v == 9 || v == 10 || (v >= 17 && v <= 23) || v == 25 || (v >= 27 && v <= 29) || v == 62 || v == 193 || v == 225))
        return fail(VL2_E_UNSUPP, "vl2_gemm: variant %d is a lab form: built into libvl2hip_lab.so only (scripts/build_lab_lib.sh)", v);
    GemmCtl ctl{d->ws, d->ws_bytes, v, (d->flags & VL2_GEMM_SPLITK) != 0};
    ctl.persist = (d->flags & VL2_GEMM_PERSISTENT) != 0;
    ctl.no_mix = (d->flags & VL2_GEMM_NO_MIX) != 0;
    ctl.no_fill = (d->flags & VL2_GEMM_NO_FILL) != 0;
    ctl.weave = (d->flags & VL2_GEMM_WEAVE) != 0;
    ctl.weave4 = (d->flags & VL2_GEMM_WEAVE4) != 0;
    ctl.no_weave4 = (d->flags & VL2_GEMM_NO_WEAVE4) != 0;
    {   // the 16 x 16 x 32 kernel: plain rows, bf16 output, no activation, no statistics out; the flag is a wish (ignored where the kernel is not built), variant 16 a demand
        const bool ok16 = !g && !f32 && !remap && act == VL2_ACT_NONE && N % 256 == 0 && !(d->stats_out && (d->flags & VL2_GEMM_SWIGLU));
        if (((v >= 16 && v <= 23) || (v >= 25 && v <= 29)) && !ok16) return fail(VL2_E_UNSUPP, "vl2_gemm: variant 16 (16x16x32 MFMA) is built for plain bf16 outputs without activation / gather / remap / stats_out, N %% 256 == 0");
        ctl.mfma16 = ok16 && ((v >= 16 && v <= 23) || (v >= 25 && v <= 29) ||
                              ((v == 0 || ((v == 256 || v == 224 || v == 192) && !(d->flags & VL2_GEMM_SWIGLU))) && (d->flags & VL2_GEMM_MFMA16)));     // 17 ... 22: lab forms (k_gemm9.h MODE 1 ... 6)
    }
    bool need_fin = d->row_norm_out && (d->flags & VL2_GEMM_NO_TICKET);      // A/B: the separate launch as in rounds 3-4
    ctl.fin = &need_fin;
    GemmArgs a{};
    a.A = (const bf16_t*)d->A; a.W = (const bf16_t*)d->W; a.C = d->C; a.bias = d->bias; a.res = (const bf16_t*)d->res;
    a.a_idx = d->a_idx; a.zero_row = nullptr;
    a.M = M; a.N = N; a.K = K; a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres; a.seg_k = d->seg_k;
    a.out_grp = d->out_grp; a.out_grp_pad = d->out_grp_pad; a.out_row_off = d->out_row_off;
    a.res_row_mod = d->res_row_mod; a.res_row_off = d->res_row_off;
    a.tiles_m = (M + GEMM_BM - 1) / GEMM_BM; a.tiles_n = N / GEMM_BN;
    a.idx_ld = M;
    a.stats_out = d->stats_out; a.stats_out_np = N / 64;
    a.stats_in = d->stats_in; a.stats_in_np = K / 64;
    a.norm = d->norm; a.norm_eps = d->norm_eps; a.w_colsum = d->w_colsum; a.row_norm = d->row_norm;
    if (d->row_norm_out && !need_fin) { a.row_norm_out = d->row_norm_out; a.row_ticket = (unsigned*)d->row_ticket; a.norm_out = d->norm_out; a.norm_out_eps = d->norm_out_eps; }
    a.tile_ctr = d->tile_ctr ? (unsigned*)d->tile_ctr : d->ws ? (unsigned*)((char*)d->ws + GEMM6_CTR_OFF) : nullptr;
    if (d->tile_ctr && ((uintptr_t)d->tile_ctr & 7)) return fail(VL2_E_BADARG, "vl2_gemm: tile_ctr must be 8-byte aligned");
    hipStream_t s = ST(stream);
    // The kernels address A and W through raw buffer resources: 32-bit byte offsets, NUM_RECORDS 2^31 - 1.  Operands beyond
    // that are covered in chunks of rows (A, C, residual, statistics) / columns (W, bias, w_colsum, C columns): e.g. the
    // [152064 x 8192] bf16 lm_head of VideoLLaMA2-72B (2.49 GB) runs as two column chunks.  Chunk sizes are multiples of
    // 256 (every tile shape, the 64-row SwiGLU blocks).  The gathered row pool cannot be chunked (its extent is unknown here).
    const int64_t LIM = ((int64_t)1 << 31) - 65536;
    int64_t rows_a = M, rows_w = N;
    if (!g && ((int64_t)(M - 1) * d->lda + K) * 2 >= LIM) rows_a = ((LIM / 2 - K) / d->lda + 1) / 256 * 256;
    if (((int64_t)(N - 1) * d->ldw + K) * 2 >= LIM) rows_w = ((LIM / 2 - K) / d->ldw + 1) / 256 * 256;
    if (rows_a <= 0 || rows_w <= 0) return fail(VL2_E_SHAPE, "vl2_gemm: a single 256-row block of A or W exceeds 2 GiB (lda %d, ldw %d)", d->lda, d->ldw);
    if ((rows_a < M || rows_w < N) && remap) return fail(VL2_E_UNSUPP, "vl2_gemm: row remap with >= 2 GiB operands not built");
    for (int64_t m0 = 0; m0 < M; m0 += rows_a) {
        const int mr = (int)(M - m0 < rows_a ? M - m0 : rows_a);
        for (int64_t n0 = 0; n0 < N; n0 += rows_w) {
            const int nr = (int)(N - n0 < rows_w ? N - n0 : rows_w);
            GemmArgs c = (mr == M) ? a : gemm_rows(a, (int)m0, mr, f32);
            if (nr != N) {
                c.N = nr;
                c.tiles_n = nr / GEMM_BN;
                c.W = a.W + (size_t)n0 * a.ldw;
                if (a.bias) c.bias = a.bias + n0;
                if (a.w_colsum) c.w_colsum = a.w_colsum + n0;
                const size_t ccol = sw ? (size_t)n0 / 2 : (size_t)n0;
                c.C = f32 ? (void*)((float*)c.C + ccol) : (void*)((bf16_t*)c.C + ccol);
                if (c.res) c.res = c.res + ccol;
                if (c.stats_out) c.stats_out = c.stats_out + (size_t)(n0 / 64) * 2;
                if (c.row_norm_out) { c.row_norm_out = nullptr; need_fin = true; }      // a column chunk sees a part of the row's partials only
            }
            const int32_t rc = gemm_dispatch(c, ctl, act, sw, f32, s);
            if (rc) return rc;
        }
    }
    if (need_fin)      // a kernel without the producer-side finalize ran (64 x 64 tiles, split-K, column chunks), or VL2_GEMM_NO_TICKET asked for the launch
        hipLaunchKernelGGL(row_norm_finalize_kernel, dim3((M + 31) / 32), dim3(256), 0, s, (const float*)d->stats_out, d->row_norm_out, M, N / 64, N, d->norm_out, d->norm_out_eps);
    return launched("vl2_gemm");
}

extern "C" int32_t vl2_row_norm_finalize(const float* stats, float* row_norm, int32_t rows, int32_t np, int32_t K, int32_t norm, float eps,
                                         void* stream) {
    if (!stats || !row_norm || rows <= 0 || np <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_row_norm_finalize: null pointer or empty shape");
    if (norm != VL2_NORM_RMS && norm != VL2_NORM_LN) return fail(VL2_E_BADARG, "vl2_row_norm_finalize: unknown norm %d", norm);
    if (!ALIGNED16(stats) || (((uintptr_t)row_norm) & 7)) return fail(VL2_E_SHAPE, "vl2_row_norm_finalize: stats must be 16-byte, row_norm 8-byte aligned");
    hipLaunchKernelGGL(row_norm_finalize_kernel, dim3((rows + 31) / 32), dim3(256), 0, ST(stream), stats, row_norm, rows, np, K, norm, eps);
    return launched("vl2_row_norm_finalize");
}

extern "C" int32_t vl2_row_stats(const void* x, float* stats, int32_t rows, int32_t C, int32_t ldx, void* stream) {
    if (!x || !stats || rows <= 0 || C <= 0) return fail(VL2_E_BADARG, "vl2_row_stats: null pointer or empty shape");
    if (C % 64 || ldx % 8 || !ALIGNED16(x)) return fail(VL2_E_SHAPE, "vl2_row_stats: need C%%64==0 and 16-byte aligned rows (C=%d)", C);
    hipLaunchKernelGGL(row_stats_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST(stream), (const bf16_t*)x, stats, rows, C, ldx);
    return launched("vl2_row_stats");
}

// ------------------------------------------------------------------------------------------------ skinny-M GEMM (batched decode)
template <int MT>
static void launch_skinny(const SkinnyArgs& a, int ks, size_t lds, hipStream_t s) {
    lds_attr<gemm_skinny_kernel<MT>>(65536);
    hipLaunchKernelGGL((gemm_skinny_kernel<MT>), dim3(a.N / 64, ks), dim3(256), lds, s, a);
}
extern "C" int32_t vl2_gemm_skinny_bf16(const void* A, const void* W, void* C, const float* bias, const void* res, int32_t M,
                                        int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t ldres, int32_t flags,
                                        void* ws, int64_t ws_bytes, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_gemm_skinny_bf16: null pointer or empty shape");
    const bool sw = flags & VL2_GEMM_SWIGLU, f32 = flags & VL2_GEMM_OUT_F32;
    if (M > 64 || N % 64 || (sw && N % 128) || K % 32 || lda % 8 || ldw % 8 || ldc % 4 || (res && ldres % 4))
        return fail(VL2_E_SHAPE, "vl2_gemm_skinny_bf16: need M<=64, N%%64==0, K%%32==0 (M=%d N=%d K=%d)", M, N, K);
    if (sw && (bias || f32)) return fail(VL2_E_UNSUPP, "vl2_gemm_skinny_bf16: SWIGLU excludes bias / f32 output");
    if (!ws || !ALIGNED16(ws) || ws_bytes <= 0) return fail(VL2_E_BADARG, "vl2_gemm_skinny_bf16: needs a 16-byte aligned workspace (fp32 partial sums)");
    const int mt = M <= 16 ? 1 : M <= 32 ? 2 : 4, Mp = 16 * mt;
    const int steps = K / 32;
    // K split: enough (column group, K slice) waves to keep ~4096 in flight, a divisor of the 32-deep steps, partials within
    // the workspace
    int ks = (4096 + N / 16 - 1) / (N / 16);
    ks = ks < 1 ? 1 : ks > 32 ? 32 : ks;
    while (ks > 1 && (steps % ks || (int64_t)ks * Mp * N * 4 > ws_bytes)) --ks;
    if ((int64_t)ks * Mp * N * 4 > ws_bytes) return fail(VL2_E_BADARG, "vl2_gemm_skinny_bf16: workspace too small (%lld bytes)", (long long)ws_bytes);
    const int kslice = K / ks;
    int kchunk = kslice;                                      // largest 32-multiple divisor of the slice whose x chunk fits 64 KiB
    while (kchunk > 32 && (kslice % kchunk || kchunk % 32 || (size_t)Mp * (kchunk + 8) * 2 > 65536)) kchunk -= 32;
    if (kslice % kchunk || (size_t)Mp * (kchunk + 8) * 2 > 65536) return fail(VL2_E_SHAPE, "vl2_gemm_skinny_bf16: no K chunking for K=%d", K);
    SkinnyArgs a{(const bf16_t*)A, (const bf16_t*)W, (float*)ws, M, N, K, lda, ldw, kslice, kchunk};
    const size_t lds = (size_t)Mp * (kchunk + 8) * 2;
    hipStream_t s = ST(stream);
    if (mt == 1) launch_skinny<1>(a, ks, lds, s); else if (mt == 2) launch_skinny<2>(a, ks, lds, s); else launch_skinny<4>(a, ks, lds, s);
    SkinnyReduceArgs r{(const float*)ws, C, bias, (const bf16_t*)res, M, Mp, N, ks, ldc, ldres};
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
                                   float rescale, float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b,
                                   void* stream) {
    if (!frames_thwc || !out || T <= 0 || std_r == 0.f || std_g == 0.f || std_b == 0.f) return fail(VL2_E_BADARG, "vl2_patchify_u8: null pointer, empty shape or zero std");
    if (G * P > H || G * P > W || Kp % 8 || Kp < 3 * P * P) return fail(VL2_E_SHAPE, "vl2_patchify_u8: bad geometry");
    U8Norm n{rescale, {mean_r, mean_g, mean_b}, {1.0f / std_r, 1.0f / std_g, 1.0f / std_b}};
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
                                int32_t group, float scale, int32_t causal, int32_t causal_off, int32_t D, int32_t variant,
                                void* stream) {
    if (!q || !k || !v || !o || B <= 0 || H <= 0 || nq <= 0 || nk <= 0 || group <= 0)
        return fail(VL2_E_BADARG, "vl2_attn_fwd: null pointer or empty shape");
    if ((q_rs | k_rs | v_rs | o_rs) % 8 || (q_bs | q_hs | k_bs | k_hs | v_bs | v_hs | o_bs | o_hs) % 4 || !ALIGNED16(q) ||
        !ALIGNED16(k) || !ALIGNED16(v) || ((uintptr_t)o & 7))
        return fail(VL2_E_SHAPE, "vl2_attn_fwd: strides must keep 16-byte row alignment");
    if (causal && causal_off < 0) return fail(VL2_E_SHAPE, "vl2_attn_fwd: causal_off must be >= 0");
    if (variant < 0 || variant > 5) return fail(VL2_E_BADARG, "vl2_attn_fwd: unknown variant %d", variant);
    const bool cls_shape = D == 64 && !causal && nq == nk && nk > 64 && (nk - 1) % 64 == 0 && group == 1;
    if (variant == 5 && (!kLab || !((D == 128 && causal) || cls_shape)))
        return fail(VL2_E_UNSUPP, "vl2_attn_fwd: variant 5 (the default kernels without round 6's scheduling changes) is a lab form of the causal head_dim 128 and class-token kernels: libvl2hip_lab.so only");
    // auto: the LDS-DMA / transpose-read structure wherever it is built (measured on MI355X, profiles/r02_attn_ab_*.jsonl:
    // causal D=128 S=945 / 1621 / 2973: 23.3 / 38.1 / 97.7 us vs 26.3 / 43.8 / 106.1 us; ViT D=64 T=8 / 16 / 32: 25.4 / 43.0 / 78.5 vs
    // 26.9 / 43.8 / 78.1 us); head_dim 96 (SigLIP's padded 72) stays on the register-staged kernel
    // full attention over [class token | 64 n patch tokens] (the CLIP tower: 577 = 1 + 576): the class token is peeled off the tiling
    // (k_attn2.h CLS = true: nine key tiles instead of ten, no dead query rows) -- automatic choice only, variant 3 keeps the plain tiling
    const bool cls_peel = variant == 0 && D == 64 && !causal && nq == nk && nk > 64 && (nk - 1) % 64 == 0 && group == 1;
    const bool cls_peel4 = variant == 4 && D == 64 && !causal && nq == nk && nk > 64 && (nk - 1) % 64 == 0 && group == 1;
    // causal head_dim 128 (the decoders' prefill): two key streams per query block (k_attn2.h NS = 2) while a SEQUENCE has at most 352 (query block, head)
    // pairs -- the longest query block's chain of dependent tiles is the launch's makespan there, and two streams halve it.  Decided per sequence (B is left
    // out) so that a prompt gets the same bits prefilled alone or in a batch.  Measured (profiles/r06_attn_ns2_ab*.jsonl, one / two streams, us): S = 512
    // 15.7 / 14.8, 945 23.5 / 21.2, 1152 28.7 / 25.2, 1408 33.1 / 30.6, 1452 x 28 heads 32.9 / 30.5, a 512-row chunk against 4096 keys 78.7 / 66.5;
    // beyond: S = 1621 (416 pairs) 37.4 / 41.1, 2973 92.2 / 102.5 -- one 128-KiB workgroup per CU balances worse than two of 64 KiB once the CUs hold
    // more than ~1.4 of them.  (The price of the per-sequence rule: a batch of short prompts, 2 x S = 945: 27.3 / 35.6.)  The ViT form stays on one stream
    // (T = 16: 40.8 / 42.8).
    if (variant == 0 && D == 128 && causal && (long)((nq + 127) / 128) * H <= 352) variant = 4;
    if (variant == 0 && (D == 64 || D == 128)) variant = 3;
    // K / V tiles are fetched through raw buffer resources whose byte offsets and NUM_RECORDS are 32-bit
    if (((int64_t)(nk - 1) * k_rs + D) * 2 >= (int64_t)1 << 31 || ((int64_t)(nk - 1) * v_rs + D) * 2 >= (int64_t)1 << 31)
        return fail(VL2_E_SHAPE, "vl2_attn_fwd: one head's K or V rows span >= 2 GiB (nk %d, row strides %d / %d elements)", nk, k_rs, v_rs);
    AttnArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, q_bs, q_hs, q_rs, k_bs, k_hs, k_rs,
               v_bs, v_hs, v_rs, o_bs, o_hs, o_rs, nq, nk, group, H, B, scale * 1.4426950408889634f, causal_off};
    dim3 g((nq + 127) / 128, H, B), b(256);
    if (causal) g = dim3(((nq + 127) / 128) * H * B, 1, 1);
    hipStream_t s = ST(stream);
    if (variant == 4) {                                   // k_attn2.h with two key streams per query block (NS = 2): 512 threads, four LDS stages
        if (D != 64 && D != 128) return fail(VL2_E_SHAPE, "vl2_attn_fwd: variant 4 is built for head_dim 64 and 128 (got %d)", D);
        const dim3 b2(512);
        const int lds_bytes = 4 * (2 * 64 * D * 2);
#define VL2_ATTN_NS2(KERN, GRID) do { lds_attr<KERN>(lds_bytes); hipLaunchKernelGGL(KERN, GRID, b2, lds_bytes, s, a); } while (0)
        if (cls_peel4) VL2_ATTN_NS2((attn2_fwd_kernel<64, false, true, 2>), dim3((nq - 1 + 127) / 128 + ((((nq - 1) & 127) == 0 || ((nq - 1) & 127) > 96) ? 1 : 0), H, B));
        else if (D == 64 && !causal) VL2_ATTN_NS2((attn2_fwd_kernel<64, false, false, 2>), g);
        else if (D == 64 && causal) VL2_ATTN_NS2((attn2_fwd_kernel<64, true, false, 2>), g);
        else if (D == 128 && !causal) VL2_ATTN_NS2((attn2_fwd_kernel<128, false, false, 2>), g);
        else VL2_ATTN_NS2((attn2_fwd_kernel<128, true, false, 2>), g);
#undef VL2_ATTN_NS2
        return launched("vl2_attn_fwd");
    }
#ifdef VL2_LAB
    if (variant == 5) {
        if (cls_shape) hipLaunchKernelGGL((attn2_fwd_kernel<64, false, true, 1, false>), dim3((nq - 1 + 127) / 128 + ((((nq - 1) & 127) == 0 || ((nq - 1) & 127) > 96) ? 1 : 0), H, B), b, 0, s, a);
        else hipLaunchKernelGGL((attn2_fwd_kernel<128, true, false, 1, false>), g, b, 0, s, a);
        return launched("vl2_attn_fwd");
    }
#endif
    if (variant == 3) {                                   // second structure (k_attn2.h): LDS-DMA ring + transpose reads
        if (cls_peel) hipLaunchKernelGGL((attn2_fwd_kernel<64, false, true>), dim3(H, B, (nq - 1 + 127) / 128 + ((((nq - 1) & 127) == 0 || ((nq - 1) & 127) > 96) ? 1 : 0)), b, 0, s, a);      // (query block = the slowest index: k_attn2.h)
        else if (D == 64 && !causal) hipLaunchKernelGGL((attn2_fwd_kernel<64, false>), g, b, 0, s, a);
        else if (D == 64 && causal) hipLaunchKernelGGL((attn2_fwd_kernel<64, true>), g, b, 0, s, a);
        else if (D == 128 && !causal) hipLaunchKernelGGL((attn2_fwd_kernel<128, false>), g, b, 0, s, a);
        else if (D == 128 && causal) hipLaunchKernelGGL((attn2_fwd_kernel<128, true>), g, b, 0, s, a);
        else return fail(VL2_E_SHAPE, "vl2_attn_fwd: variant 3 is built for head_dim 64 and 128 (got %d)", D);
        return launched("vl2_attn_fwd");
    }
    if (D == 64 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<64, false>), g, b, 0, s, a);
    else if (D == 64 && causal) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), g, b, 0, s, a);
    else if (D == 128 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<128, false>), g, b, 0, s, a);
    else if (D == 128 && causal) {
        // two KV groups per workgroup when a sequence has few (q block, head) pairs: decided per SEQUENCE (B is left out) so
        // that a prompt gets the same bits prefilled alone or in a batch.  Measured (scripts/attn_bench.py): 256 pairs (S = 945,
        // 32 heads) 28.5 -> 27.1 us, 336 (S = 1452, 28 heads) 38.9 -> 35.8; 416 (S = 1621, 32 heads) 47.4 -> 52.2: the SIMD's
        // per-tile throughput, not the length of the dependent tile chain, is the limit once every CU has > 1.4 workgroups.
        const long per_seq = (long)((nq + 127) / 128) * H;
        const bool two = variant == 2 || (variant == 0 && per_seq <= 352);
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
// strip form (k_stc.h): TPF teams per frame, a function of the grid (H, W) alone so that a frame's partial-sum grouping never depends on F
#define DWS_TPF_MAX 96
static inline int dws_tpf(int H, int W) {
    const int U = H * ((W + DWS_P - 1) / DWS_P);
    return U <= DWS_TPF_MAX ? U : (U + (U + 63) / 64 - 1) / ((U + 63) / 64);
}
template <int NVT>
static void launch_dwconv_strip(const void* x, void* y, const float* w9c, const float* lnw, const float* lnb, float* psum, int F, int H, int W, int C,
                                float eps, hipStream_t s) {
    const int U = H * ((W + DWS_P - 1) / DWS_P), tpf = dws_tpf(H, W), iters = (U + tpf - 1) / tpf;
    const size_t lds = (size_t)9 * C * 2;
    lds_attr<dwconv_strip_ln_silu_kernel<NVT>>((int)lds);
    hipLaunchKernelGGL((dwconv_strip_ln_silu_kernel<NVT>), dim3((F * tpf + 1) / 2), dim3(512), lds, s, (const bf16_t*)x, (bf16_t*)y, w9c, lnw, lnb, psum,
                       F, H, W, C, eps, tpf, iters);
}
extern "C" int64_t vl2_dwconv_mean_workspace_bytes(int32_t F, int32_t C) {
    if (F <= 0 || C <= 0) return -1;
    return (int64_t)F * DWS_TPF_MAX * C * 4;
}
extern "C" int32_t vl2_dwconv3x3_ln_silu_mean(const void* x, void* y, const float* w9c, const float* lnw, const float* lnb, int32_t F, int32_t H,
                                              int32_t W, int32_t C, float eps, float* mean, void* ws, int64_t ws_bytes, void* stream) {
    if (!x || !y || !w9c || !lnw || !lnb || F <= 0 || H <= 0 || W <= 0) return fail(VL2_E_BADARG, "vl2_dwconv3x3_ln_silu_mean: bad args");
    if (C % 8 || C > 8192) return fail(VL2_E_SHAPE, "vl2_dwconv3x3_ln_silu_mean: need C%%8==0 and C<=8192");
    const int tpf = dws_tpf(H, W);
    if (mean && (!ws || !ALIGNED16(ws) || ws_bytes < (int64_t)F * tpf * C * 4))
        return fail(VL2_E_BADARG, "vl2_dwconv3x3_ln_silu_mean: workspace missing, unaligned or < %lld bytes", (long long)F * tpf * C * 4);
    float* psum = mean ? (float*)ws : nullptr;
    if (C <= 2048) launch_dwconv_strip<1>(x, y, w9c, lnw, lnb, psum, F, H, W, C, eps, ST(stream));
    else if (C <= 4096) launch_dwconv_strip<2>(x, y, w9c, lnw, lnb, psum, F, H, W, C, eps, ST(stream));
    else launch_dwconv_strip<4>(x, y, w9c, lnw, lnb, psum, F, H, W, C, eps, ST(stream));
    if (mean)
        hipLaunchKernelGGL(chan_psum_finish_kernel, dim3((C + 255) / 256, F), dim3(256), 0, ST(stream), (const float*)psum, mean, tpf, C, 1.0f / (float)(H * W));
    return launched("vl2_dwconv3x3_ln_silu_mean");
}
extern "C" int32_t vl2_se_excite_scale(void* x, const float* g1, const void* W2, const float* b2, int32_t F, int32_t HW, int32_t C, int32_t rd,
                                       void* stream) {
    if (!x || !g1 || !W2 || F <= 0 || HW <= 0 || C <= 0 || rd <= 0) return fail(VL2_E_BADARG, "vl2_se_excite_scale: bad args");
    if (C % 8 || rd % 16) return fail(VL2_E_SHAPE, "vl2_se_excite_scale: need C%%8==0 and rd%%16==0 (C=%d rd=%d)", C, rd);
    hipLaunchKernelGGL(se_excite_scale_kernel, dim3((C + 127) / 128, F), dim3(256), 0, ST(stream), (bf16_t*)x, g1, (const bf16_t*)W2, b2, HW, C, rd);
    return launched("vl2_se_excite_scale");
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
    hipLaunchKernelGGL(small_linear_kernel, dim3((N + SL_NB - 1) / SL_NB, (F + 7) / 8), dim3(256), 0, ST(stream), x, (const bf16_t*)W, b, out, F, N, K, a);
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

// ------------------------------------------------------------------------------------------------ one-time weight re-layout
extern "C" int32_t vl2_pack_fold_norm(const void* W, const void* g, const void* beta, const void* bias, void* Wp, float* colsum, float* shift,
                                      int32_t N, int32_t K, int32_t ldw, void* stream) {
    if (!W || !g || !Wp || !colsum || N <= 0 || K <= 0 || ldw < K) return fail(VL2_E_BADARG, "vl2_pack_fold_norm: bad args");
    if ((beta != nullptr) != (shift != nullptr)) return fail(VL2_E_BADARG, "vl2_pack_fold_norm: beta and shift go together");
    if (bias && !beta) return fail(VL2_E_BADARG, "vl2_pack_fold_norm: a bias without a norm shift needs no folding");
    hipLaunchKernelGGL(pack_fold_norm_kernel, dim3((N + 3) / 4), dim3(256), 0, ST(stream), (const bf16_t*)W, (const bf16_t*)g, (const bf16_t*)beta,
                       (const bf16_t*)bias, (bf16_t*)Wp, colsum, shift, N, K, ldw);
    return launched("vl2_pack_fold_norm");
}
extern "C" int32_t vl2_pack_gate_up(const void* gate, const void* up, void* out, int32_t I, int32_t D, void* stream) {
    if (!gate || !up || !out || I <= 0 || D <= 0) return fail(VL2_E_BADARG, "vl2_pack_gate_up: bad args");
    if (I % 32 || D % 8) return fail(VL2_E_SHAPE, "vl2_pack_gate_up: need I%%32==0 and D%%8==0");
    hipLaunchKernelGGL(pack_gate_up_kernel, dim3(2 * I), dim3(128), 0, ST(stream), (const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)out, D);
    return launched("vl2_pack_gate_up");
}
extern "C" int32_t vl2_pack_permute(const void* in, void* out, int32_t A, int32_t B, int32_t C, int32_t out_f32, void* stream) {
    if (!in || !out || A <= 0 || B <= 0 || C <= 0 || A > 65535) return fail(VL2_E_BADARG, "vl2_pack_permute: bad args");
    if ((int64_t)B * C > 0x7fffffffLL) return fail(VL2_E_SHAPE, "vl2_pack_permute: plane too large");
    const dim3 grid((unsigned)(((int64_t)B * C + 255) / 256), A);
    if (out_f32) hipLaunchKernelGGL((pack_permute_kernel<true>), grid, dim3(256), 0, ST(stream), (const bf16_t*)in, out, B, C);
    else hipLaunchKernelGGL((pack_permute_kernel<false>), grid, dim3(256), 0, ST(stream), (const bf16_t*)in, out, B, C);
    return launched("vl2_pack_permute");
}
extern "C" int32_t vl2_pack_pad_rows(const void* in, void* out, int64_t rows, int64_t cols_src, int64_t cols_dst, void* stream) {
    if (!in || !out || rows <= 0 || cols_src <= 0 || cols_dst < cols_src || rows > 0x7fffffffLL) return fail(VL2_E_BADARG, "vl2_pack_pad_rows: bad args");
    hipLaunchKernelGGL(pack_pad_rows_kernel, dim3((unsigned)rows), dim3(256), 0, ST(stream), (const bf16_t*)in, (bf16_t*)out, (long)cols_src, (long)cols_dst);
    return launched("vl2_pack_pad_rows");
}
extern "C" int32_t vl2_pack_cvt_f32(const void* in, float* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0) return fail(VL2_E_BADARG, "vl2_pack_cvt_f32: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_cvt_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), (const bf16_t*)in, out, (long)n);
    return launched("vl2_pack_cvt_f32");
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
    // one output row per wave: measured on MI355X 3.09 / 3.30 / 3.97 ms per 7B decode token at 1 / 2 / 4 rows per wave
    // norm-carrying single-pass rows (q/k/v, lm_head): x requested before the weight row (k_decode.h gemv_xfirst_bf16_kernel, same bits)
    if constexpr (!SW) {
        if ((a.norm_w || a.rms_plain) && a.K <= 4096) {
            hipLaunchKernelGGL((gemv_xfirst_bf16_kernel<false, F32, 2>), dim3((n_out + 3) / 4), dim3(256), (size_t)a.K * 2, s, a);
            return;
        }
    }
    hipLaunchKernelGGL((gemv_bf16_kernel<SW, F32, 1>), dim3((n_out + 3) / 4), dim3(256), (size_t)a.K * 2, s, a);
}
extern "C" int32_t vl2_gemv_bf16(const void* W, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                                 int32_t N, int32_t K, int32_t ldw, float eps, int32_t flags, void* stream) {
    if (!W || !x || !y || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_gemv_bf16: bad args");
    if (K % 8 || ldw % 8 || K > 32704) return fail(VL2_E_SHAPE, "vl2_gemv_bf16: need K%%8==0, K<=32704 (x lives in LDS; K=%d)", K);
    const bool sw = flags & VL2_GEMM_SWIGLU, f32 = flags & VL2_GEMM_OUT_F32;
    if (sw && (N % 64 || f32 || bias)) return fail(VL2_E_SHAPE, "vl2_gemv_bf16: SWIGLU needs N%%64==0, bf16 output, no bias");
    GemvArgs a{(const bf16_t*)W, (const bf16_t*)x, norm_w, (const bf16_t*)res, y, N, K, ldw, eps, bias, 0, 0, 0, 0};
    if (flags & VL2_GEMV_RMS_PLAIN) { a.norm_w = nullptr; a.rms_plain = 1; }
    if (sw) launch_gemv<true, false>(a, N / 2, ST(stream));
    else if (f32) launch_gemv<false, true>(a, N, ST(stream));
    else launch_gemv<false, false>(a, N, ST(stream));
    return launched("vl2_gemv_bf16");
}
extern "C" int32_t vl2_pack_quant_fp8(const void* w, int64_t N, int64_t K, int64_t ldw, void* q, float* scale, void* stream) {
    if (!w || !q || !scale || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_pack_quant_fp8: bad args");
    if (K % 16 || ldw % 8 || ldw < K || N > 0x7fffffff || K > 0x7fffffff) return fail(VL2_E_SHAPE, "vl2_pack_quant_fp8: need K%%16==0, ldw%%8==0, ldw>=K");
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)N), dim3(256), 0, ST(stream), (const bf16_t*)w, (uint8_t*)q, scale, (int)K, (long)ldw);
    return launched("vl2_pack_quant_fp8");
}
extern "C" int32_t vl2_quant_act_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* row_tab, int32_t M, int32_t K, int32_t norm, float eps, void* stream) {
    if (!x || !q || !row_tab || M <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_quant_act_fp8: bad args");
    if (norm != VL2_NORM_NONE && norm != VL2_NORM_RMS) return fail(VL2_E_BADARG, "vl2_quant_act_fp8: norm must be none or RMS");
    if (K % 16 || ldx % 8 || ldq % 16 || ldx < K || ldq < K || !ALIGNED16(x) || !ALIGNED16(q) || (((uintptr_t)row_tab) & 7))
        return fail(VL2_E_SHAPE, "vl2_quant_act_fp8: need K%%16==0, 16-byte aligned rows");
    hipLaunchKernelGGL(quant_act_fp8_kernel, dim3((unsigned)M), dim3(256), 0, ST(stream), (const bf16_t*)x, (long)ldx, (uint8_t*)q, (long)ldq, row_tab, K,
                       norm == VL2_NORM_RMS ? 1 : 0, eps);
    return launched("vl2_quant_act_fp8");
}
extern "C" int32_t vl2_gemv_fp8(const void* q, const float* scale, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                                int32_t N, int32_t K, int32_t ldq, float eps, int32_t flags, void* stream) {
    if (!q || !scale || !x || !y || N <= 0 || K <= 0) return fail(VL2_E_BADARG, "vl2_gemv_fp8: bad args");
    if (K % 16 || ldq % 16 || ldq < K || K > 32704 || N % 2) return fail(VL2_E_SHAPE, "vl2_gemv_fp8: need N even, K%%16==0, ldq%%16==0, K<=32704 (x lives in LDS as 16-bit elements; K=%d)", K);
    const bool sw = flags & VL2_GEMM_SWIGLU, f32 = flags & VL2_GEMM_OUT_F32;
    if (sw && (N % 64 || f32 || bias)) return fail(VL2_E_SHAPE, "vl2_gemv_fp8: SWIGLU needs N%%64==0, 16-bit output, no bias");
    Gemv8Args a{(const uint8_t*)q, scale, (const bf16_t*)x, norm_w, (const bf16_t*)res, y, N, K, ldq, eps, bias, 0};
    if (flags & VL2_GEMV_RMS_PLAIN) { a.norm_w = nullptr; a.rms_plain = 1; }
    const size_t lds = (size_t)K * 2;
    const dim3 b(256);
    // rows of K <= 4096 are four 16-B vectors per lane: two pairs per wave and trip keep the 16-bit kernel's bytes in flight (k_fp8.h)
#define VL2_G8(SW, F32, NP) hipLaunchKernelGGL((gemv_fp8_kernel<SW, F32, NP>), dim3((unsigned)((N / 2 + 4 * NP - 1) / (4 * NP))), b, lds, ST(stream), a)
    if (K <= 4096) { if (sw) VL2_G8(true, false, 2); else if (f32) VL2_G8(false, true, 2); else VL2_G8(false, false, 2); }
    else           { if (sw) VL2_G8(true, false, 1); else if (f32) VL2_G8(false, true, 1); else VL2_G8(false, false, 1); }
#undef VL2_G8
    return launched("vl2_gemv_fp8");
}
template <bool SW, bool F32>
static void launch_gemv_mr(const GemvArgs& a, int mb, int n_out, hipStream_t s) {
    // two output rows per wave amortise staging the MB rows of x: measured at B = 4, 5.56 / 5.04 / 5.56 ms per step at 1 / 2 / 4
    constexpr int rpw = 2;
    const dim3 g((n_out + 4 * rpw - 1) / (4 * rpw)), b(256);
    const size_t lds = (size_t)mb * a.K * 2;
#define VL2_MR(MBV) hipLaunchKernelGGL((gemv_mr_bf16_kernel<SW, F32, MBV, 2>), g, b, lds, s, a)
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
    hipLaunchKernelGGL(attn_decode_kernel<false>, dim3(nsplit, nkv, (group + 3) / 4), dim3(256), 0, ST(stream), (const bf16_t*)qkv, (bf16_t*)kcache,
                       (bf16_t*)vcache, cos_t, sin_t, partial, nh, group, nkv, smax, pos, pos_dev, scale * 1.4426950408889634f, 0L, 0L, 0L,
                       (int*)nullptr, (bf16_t*)nullptr);
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(nh), dim3(128), 0, ST(stream), partial, (bf16_t*)out, nsplit, pos, pos_dev, 0L, 0L);
    return launched("vl2_attn_decode");
}
#ifdef VL2_LAB
// attention + combine of ONE decode token in one launch (k_decode.h attn_decode_kernel<true>): `cnt` = nkv int32 ticket counters,
// zero when the launch starts (vl2_llm_decode_step clears the counters of all layers in its argmax launch).  Same bits as
// vl2_attn_decode.  Measured 2.4 us per layer SLOWER than the two launches (profiles/r03_experiments.md section 5), so the stage-level
// decode step takes it only with VL2_STAGE_FUSED_DECODE_ATTN in the descriptor flags; exported as vl2_attn_decode_fused and kept under test.
static int32_t attn_decode_fused(const void* qkv, void* kcache, void* vcache, const float* cos_t, const float* sin_t, float* partial, void* out,
                                 int32_t nh, int32_t nkv, int32_t smax, const int32_t* pos_dev, float scale, int32_t* cnt, void* stream) {
    const int group = nh / nkv, nsplit = (smax + 63) / 64;
    hipLaunchKernelGGL(attn_decode_kernel<true>, dim3(nsplit, nkv, (group + 3) / 4), dim3(256), 0, ST(stream), (const bf16_t*)qkv, (bf16_t*)kcache,
                       (bf16_t*)vcache, cos_t, sin_t, partial, nh, group, nkv, smax, 0, pos_dev, scale * 1.4426950408889634f, 0L, 0L, 0L,
                       cnt, (bf16_t*)out);
    return launched("vl2_llm_decode_step (attention)");
}
extern "C" int32_t vl2_attn_decode_fused(const void* qkv, void* kcache, void* vcache, const float* cos_t, const float* sin_t, float* partial,
                                         void* out, int32_t nh, int32_t nkv, int32_t smax, const int32_t* pos_dev, float scale, int32_t* cnt,
                                         void* stream) {
    if (!qkv || !kcache || !vcache || !cos_t || !sin_t || !partial || !out || !pos_dev || !cnt || nh <= 0 || nkv <= 0 || smax <= 0 || nh % nkv)
        return fail(VL2_E_BADARG, "vl2_attn_decode_fused: bad args");
    return attn_decode_fused(qkv, kcache, vcache, cos_t, sin_t, partial, out, nh, nkv, smax, pos_dev, scale, cnt, stream);
}
#endif
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
    hipLaunchKernelGGL(attn_decode_kernel<false>, dim3(nsplit, nkv * B, (group + 3) / 4), dim3(256), 0, ST(stream), (const bf16_t*)qkv,
                       (bf16_t*)kcache, (bf16_t*)vcache, cos_t, sin_t, partial, nh, group, nkv, smax, 0, pos_dev,
                       scale * 1.4426950408889634f, (long)qkv_bs, (long)cache_bs, partial_bs, (int*)nullptr, (bf16_t*)nullptr);
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(nh, B), dim3(128), 0, ST(stream), partial, (bf16_t*)out, nsplit, 0, pos_dev,
                       partial_bs, (long)out_bs);
    return launched("vl2_attn_decode_batched");
}
extern "C" int32_t vl2_argmax(const float* logits, int32_t V, int32_t* tok, int32_t* hist, int32_t step, int32_t* state,
                              void* stream) {
    if (!logits || !tok || V <= 0) return fail(VL2_E_BADARG, "vl2_argmax: bad args");
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, ST(stream), logits, V, tok, hist, step, state, (int*)nullptr, 0, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0);
    return launched("vl2_argmax");
}
// do_sample=True: temperature -> top-k -> top-p -> one draw at the host's uniform number (k_sample.h; HF:generation/logits_process.py warpers +
// GenerationMixin._sample).  Same token / history / state protocol as vl2_argmax, so it takes the argmax launch's place in a decode loop or graph.
extern "C" int32_t vl2_sample_token(const float* logits, int32_t V, float temperature, int32_t top_k, float top_p, const float* u, int32_t* tok,
                                    int32_t* hist, int32_t step, int32_t* state, float* dbg, void* stream) {
    if (!logits || !tok || !u || V <= 0) return fail(VL2_E_BADARG, "vl2_sample_token: null pointer or empty vocabulary");
    if (!(temperature > 0.f) || top_k < 0 || !(top_p > 0.f)) return fail(VL2_E_BADARG, "vl2_sample_token: need temperature > 0, top_k >= 0, top_p > 0 (got %g, %d, %g)", (double)temperature, top_k, (double)top_p);
    if (!state && step < 0) return fail(VL2_E_BADARG, "vl2_sample_token: negative step");
    SampleArgs a{logits, V, temperature, top_k, top_p, u, tok, hist, step, state, dbg};
    if (V <= 32768) {                  // the scaled scores fit into LDS beside the kernel's 10 KB of static LDS
        lds_attr<sample_token_kernel<true>>(32768 * 4);
        hipLaunchKernelGGL(sample_token_kernel<true>, dim3(1), dim3(1024), (size_t)V * 4, ST(stream), a);
    } else {
        hipLaunchKernelGGL(sample_token_kernel<false>, dim3(1), dim3(1024), 0, ST(stream), a);
    }
    return launched("vl2_sample_token");
}
// the decode step's argmax, which also clears `nzero` int32 words (the fused attention launches' ticket counters)
static int32_t argmax_and_clear(const float* logits, int32_t V, int32_t* tok, int32_t* hist, int32_t* state, int32_t* zero, int32_t nzero,
                                const void* embed, void* x0, int32_t D, void* stream) {
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, ST(stream), logits, V, tok, hist, 0, state, (int*)zero, nzero,
                       (const bf16_t*)embed, (bf16_t*)x0, D);
    return launched("vl2_llm_decode_step: argmax");
}
#ifdef VL2_LAB
extern "C" int32_t vl2_decode_tail(const void* Wo, const void* Wgu, const void* Wd, int32_t ldwo, int32_t ldwgu, int32_t ldwd, const void* o,
                                   const void* x0, void* x1, void* act, void* xout, int32_t D, int32_t QD, int32_t I, float eps, int32_t* bar,
                                   void* stream) {
    if (!Wo || !Wgu || !Wd || !o || !x0 || !x1 || !act || !xout || !bar) return fail(VL2_E_BADARG, "vl2_decode_tail: null pointer");
    if (D <= 0 || QD <= 0 || I <= 0 || D % 8 || QD % 8 || I % 32 || ldwo % 8 || ldwgu % 8 || ldwd % 8 || D > 32512 || QD > 32512 || I > 32512)
        return fail(VL2_E_SHAPE, "vl2_decode_tail: need D, QD %% 8 == 0, I %% 32 == 0, all <= 32512 (D %d QD %d I %d)", D, QD, I);
    if ((uintptr_t)bar & 3) return fail(VL2_E_BADARG, "vl2_decode_tail: bar must be 4-byte aligned");
    TailArgs a{(const bf16_t*)Wo, (const bf16_t*)Wgu, (const bf16_t*)Wd, ldwo, ldwgu, ldwd, (const bf16_t*)o, (const bf16_t*)x0, (bf16_t*)x1,
               (bf16_t*)act, (bf16_t*)xout, D, QD, I, eps, (unsigned*)bar};
    const int kmax = QD > D ? (QD > I ? QD : I) : (D > I ? D : I);
    const int g = cu_count() & ~7;                                 // one workgroup per CU (the grid barrier needs all of them resident), 8 groups
    if (g <= 0 || (long)I > (long)TAIL_MAX_ROWS * TAIL_WAVES * g || (long)D > (long)TAIL_MAX_ROWS * TAIL_WAVES * g)
        return fail(VL2_E_SHAPE, "vl2_decode_tail: %d rows per phase exceed %d per wave on %d workgroups", I > D ? I : D, TAIL_MAX_ROWS, g);
    hipLaunchKernelGGL((decode_tail_kernel<7>), dim3(g), dim3(1024), (size_t)kmax * 2, ST(stream), a);
    return launched("vl2_decode_tail");
}
#endif
extern "C" int32_t vl2_embed_rows(const int32_t* ids, const void* table, void* out, int32_t n, int32_t D, int32_t ldo, void* stream) {
    if (!ids || !table || !out || n <= 0 || D % 8 || ldo % 8) return fail(VL2_E_BADARG, "vl2_embed_rows: bad args");
    hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(128), 0, ST(stream), ids, (const bf16_t*)table, (bf16_t*)out, D, ldo);
    return launched("vl2_embed_rows");
}

#include "vl2_stage.inc"

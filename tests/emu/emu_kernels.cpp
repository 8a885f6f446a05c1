// TEST INFRASTRUCTURE ONLY: the product's kernel headers (videollama2_amd/csrc/k_*.h) compiled for the host against
// tests/emu/hip_emu.h, exported under the SAME C ABI as libvl2hip.so (include/vl2hip.h; `stream` ignored), so the
// CPU-only test-suite can drive the real host code (videollama2_amd/ops.py, tower.py, connector.py, decoder.py,
// model.py) end to end through the real kernels' indexing logic.  Built by tests/emu/build_emu.py.
#include "hip_emu.h"
alignas(64) unsigned char vl2_smem[160 * 1024];
#include "k_gemm.h"
#include "k_gemm6.h"
#include "k_gemm7.h"
#include "k_gemm8.h"
#include "k_gemm9.h"
#include "k_norm.h"
#include "k_vit.h"
#include "k_attn.h"
#include "k_attn2.h"
#include "k_stc.h"
#include "k_decode.h"
#include "k_decode_tail.h"
#include "k_fp8.h"
#include "k_skinny.h"
#include "k_pack.h"
#include "k_sample.h"
#include <cstdint>
#include <algorithm>
#include <vector>
#define VL2_EXPERIMENTAL 1      // this translation unit DEFINES the experimental entry points too
#include "../../include/vl2hip.h"

static char g_err[256] = "emu";
extern "C" int32_t vl2_version(void) { return VL2_ABI_VERSION; }
extern "C" const char* vl2_elem_name(void) { return VL2_ELEM_NAME; }
extern "C" const char* vl2_last_error_string(void) { return g_err; }
extern "C" int64_t vl2_workspace_bytes(void) { return 64; }
extern "C" int32_t vl2_fill_zero(void* p, int64_t bytes, void*) { if (!p || bytes < 0 || (bytes & 3)) return -1; memset(p, 0, (size_t)bytes); return 0; }

// per-call controls (vl2_gemm_desc.variant, VL2_GEMM_SPLITK, vl2_attn_fwd variant): set by the entry points below
static int g_gemm_variant = 0;
static bool g_no_weave4 = false;
static int g_mfma16_mode = 0;       // variants 17 / 18: the lab orders of gemm9's LDS-DMA issue
static int g_mfma16_tile = 0;       // 256 / 224 / 192 with VL2_GEMM_MFMA16: the one-round 128 x 128 / fill-the-round tiles of the 16 x 16 x 32 set (k_gemm9.h, k_gemm7.h)
static bool g_mfma16 = false;       // variant 16 / VL2_GEMM_MFMA16: the 256 x 256 ping-pong tile on the 16 x 16 x 32 matrix instruction (k_gemm9.h)
static bool g_weave4 = false;       // VL2_GEMM_WEAVE4: the 256-/192-row ping-pong bodies with the woven LDS-DMA issue
static bool g_need_fin = false;     // a GEMM path without the producer-side finalize ran: append the row_norm_finalize launch (vl2_abi.hip GemmCtl.fin)
static bool g_gemm6_dynamic = false;       // gemm6: tiles handed out through the counter block (variants 70 / 71 = 60 / 61 dynamic)
template <int ACT, bool SW, bool F32, bool G>
static void run_gemm(GemmArgs a) {
    if constexpr (!G && !F32 && ACT == 0) {
        if constexpr (!SW) {
            if (g_mfma16 && (g_mfma16_tile == 256 || (g_mfma16_tile == 0 && g_mfma16_mode == 0 && (long)a.tiles_m * a.tiles_n <= 4))) {     // (vl2_abi.hip: <= 256 tiles; the emulator's shapes are small)
                emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm_l8_16_bf16_kernel<false>(a); });
                return;
            }
            if (g_mfma16 && (g_mfma16_tile == 224 || g_mfma16_tile == 192)) {
                const int bm = g_mfma16_tile;
                a.tiles_m = (a.M + bm - 1) / bm; a.tiles_n = a.N / 128;
                if (bm == 224) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm7_16_bf16_kernel<3>(a); });
                else emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm7_16_bf16_kernel<2>(a); });
                return;
            }
        }
        if (g_mfma16 && g_mfma16_mode == 0 && a.M > 256 && a.M % 256 != 0 && a.N % 256 == 0) {
            // vl2_abi.hip launch_gemm: a row-split call on the 16 x 16 x 32 instruction is ONE mixed launch (k_gemm9.h gemm_mix16_bf16_kernel)
            const int M1 = (a.M / 256) * 256;
            GemmArgs big = a, tail = a;
            big.M = M1; big.tiles_m = M1 / 256; big.tiles_n = a.N / 256;
            tail.M = a.M - M1; tail.A = a.A + (size_t)M1 * a.lda; tail.C = (void*)((bf16_t*)a.C + (size_t)M1 * a.ldc);
            if (a.res) tail.res = a.res + (size_t)M1 * a.ldres;
            if (a.stats_out) tail.stats_out = a.stats_out + (size_t)M1 * a.stats_out_np * 2;
            if (a.stats_in) tail.stats_in = a.stats_in + (size_t)M1 * a.stats_in_np * 2;
            if (a.row_norm) tail.row_norm = a.row_norm + (size_t)M1 * 2;
            if (a.row_norm_out) { tail.row_norm_out = a.row_norm_out + (size_t)M1 * 2; tail.row_ticket = a.row_ticket + M1 / 64; }
            tail.tiles_m = (tail.M + 127) / 128; tail.tiles_n = a.N / 128;
            const int n_big = big.tiles_m * big.tiles_n, n_all = n_big + tail.tiles_m * tail.tiles_n;
            emu::launch(dim3(n_all), dim3(512), [=] { gemm_mix16_bf16_kernel<SW>(big, tail, n_big); });
            return;
        }
        if (g_mfma16) {
            if (g_mfma16_mode != 0 && a.row_norm_out) g_need_fin = true;       // only the shipped form (MODE 0) finalizes its rows itself (vl2_abi.hip launch_gemm)
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = a.N / 256;
            if (g_mfma16_mode == 1) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 1>(a); });
            else if (g_mfma16_mode == 2) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 2>(a); });
            else if (g_mfma16_mode == 3) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 3>(a); });
            else if (g_mfma16_mode == 9) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 9>(a); });
            else if (g_mfma16_mode == 4) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 4>(a); });
            else if (g_mfma16_mode == 5) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 5>(a); });
            else if (g_mfma16_mode == 6) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 6>(a); });
            else emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm9_bf16_kernel<SW, 0>(a); });
            return;
        }
    }
    if constexpr (!SW) {                                        // fill-the-round 224 x 128 / 192 x 128 tiles (k_gemm7.h), plain and gathered
        if (g_gemm_variant == 224 || g_gemm_variant == 192 || g_gemm_variant == 225 || g_gemm_variant == 193) {   // 225 / 193: LDS-DMA issue woven into the MFMA phases
            const bool weave = (g_gemm_variant & 1) != 0;
            const int bm = g_gemm_variant & ~1;
            a.tiles_m = (a.M + bm - 1) / bm; a.tiles_n = a.N / 128;
            const dim3 grid(a.tiles_m * a.tiles_n);
            if (bm == 224) { if (weave) emu::launch(grid, dim3(512), [=] { gemm7_bf16_kernel<ACT, F32, G, 3, true>(a); }); else emu::launch(grid, dim3(512), [=] { gemm7_bf16_kernel<ACT, F32, G, 3, false>(a); }); }
            else { if (weave) emu::launch(grid, dim3(512), [=] { gemm7_bf16_kernel<ACT, F32, G, 2, true>(a); }); else emu::launch(grid, dim3(512), [=] { gemm7_bf16_kernel<ACT, F32, G, 2, false>(a); }); }
            return;
        }
    }
    if constexpr (!G) {
        if (g_gemm_variant == 4 && a.N % 256 == 0) {
            a.tiles_m = (a.M + 127) / 128; a.tiles_n = a.N / 256;
            if constexpr (!F32) {        // vl2_abi.hip want_tr_epilogue: no residual -> register-resident C^T epilogue
                if (a.res == nullptr) { emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm3_bf16_kernel<ACT, SW, false, true>(a); }); return; }
            }
            emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm3_bf16_kernel<ACT, SW, F32>(a); });
            return;
        }
        if (g_gemm_variant == 5 && a.N % 256 == 0) {            // the same with the LDS-DMA issue woven into the MFMA phases (lab form)
            a.tiles_m = (a.M + 127) / 128; a.tiles_n = a.N / 256;
            if constexpr (!F32) {
                if (a.res == nullptr) { emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm3_bf16_kernel<ACT, SW, false, true, -1, true>(a); }); return; }
            }
            emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm3_bf16_kernel<ACT, SW, F32, false, -1, true>(a); });
            return;
        }
        if (g_gemm_variant == 256) {
            emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm_l8_bf16_kernel<ACT, SW, F32>(a); });
            return;
        }
        if (g_gemm_variant == 9 && a.N % 256 == 0) {            // gemm8: the 256 x 256 tile on four waves (128 x 128 wave tiles)
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = a.N / 256;
            if constexpr (!F32) {
                if (a.res == nullptr) { emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(256), [=] { gemm8_bf16_kernel<ACT, SW, false, true>(a); }); return; }
            }
            emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(256), [=] { gemm8_bf16_kernel<ACT, SW, F32>(a); });
            return;
        }
        if (g_gemm_variant == 8 && a.N % 256 == 0) {
            a.tiles_m = (a.M + 255) / 256; a.tiles_n = a.N / 256;
            if constexpr (!F32) {
                if (g_weave4) {
                    if (a.res == nullptr) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, true, -1, 256, true>(a); });
                    else emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, false, -1, 256, true>(a); });
                    return;
                }
                if (a.res == nullptr) { emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, true>(a); }); return; }
            }
            emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, F32>(a); });
            return;
        }
    }
    if constexpr (!G && !F32) {
        if (g_gemm_variant == 24 && a.N % 256 == 0 && a.M > 256) {   // the row-split call as ONE mixed launch (vl2_abi.hip launch_gemm): big tiles on
            const int M1 = (a.M / 256) * 256 == a.M ? a.M - 256 : (a.M / 256) * 256;      // the leading whole 256-row tiles, 128x128 8-wave tiles on the rest
            GemmArgs big = a, tail = a;
            big.M = M1; big.tiles_m = M1 / 256; big.tiles_n = a.N / 256;
            tail.M = a.M - M1; tail.A = a.A + (size_t)M1 * a.lda; tail.C = (void*)((bf16_t*)a.C + (size_t)M1 * a.ldc);
            if (a.res) tail.res = a.res + (size_t)M1 * a.ldres;
            if (a.stats_out) tail.stats_out = a.stats_out + (size_t)M1 * a.stats_out_np * 2;
            if (a.stats_in) tail.stats_in = a.stats_in + (size_t)M1 * a.stats_in_np * 2;
            if (a.row_norm) tail.row_norm = a.row_norm + (size_t)M1 * 2;
            if (a.row_norm_out) { tail.row_norm_out = a.row_norm_out + (size_t)M1 * 2; tail.row_ticket = a.row_ticket + M1 / 64; }
            tail.tiles_m = (tail.M + 127) / 128; tail.tiles_n = a.N / 128;
            const int n_big = big.tiles_m * big.tiles_n, n_all = n_big + tail.tiles_m * tail.tiles_n;
            if (g_weave4) {
                if (a.res == nullptr) emu::launch(dim3(n_all), dim3(512), [=] { gemm_mix_bf16_kernel<ACT, SW, true, true>(big, tail, n_big); });
                else emu::launch(dim3(n_all), dim3(512), [=] { gemm_mix_bf16_kernel<ACT, SW, false, true>(big, tail, n_big); });
                return;
            }
            if (a.res == nullptr) emu::launch(dim3(n_all), dim3(512), [=] { gemm_mix_bf16_kernel<ACT, SW, true>(big, tail, n_big); });
            else emu::launch(dim3(n_all), dim3(512), [=] { gemm_mix_bf16_kernel<ACT, SW, false>(big, tail, n_big); });
            return;
        }
        a.tile_first_dyn = 0;
        if (g_gemm_variant == 80 || g_gemm_variant == 81) { g_gemm6_dynamic = true; a.tile_first_dyn = 1; g_gemm_variant -= 20; }
        else if (g_gemm_variant == 70 || g_gemm_variant == 71) { g_gemm6_dynamic = true; g_gemm_variant -= 10; } else g_gemm6_dynamic = false;
        if ((g_gemm_variant == 60 || g_gemm_variant == 61 || g_gemm_variant == 62) && a.N % 256 == 0 && a.res == nullptr && !a.stats_out &&
            (!a.norm || a.row_norm) && a.M >= (g_gemm_variant == 60 ? 256 : 192) && a.K >= 512) {
            // gemm6 (persistent ping-pong): 256-row / 192-row / 192-row with two accumulator sets.  THREE workgroups, so that every
            // workgroup walks several tiles (the emulator runs workgroups one after the other; a real launch has one per CU)
            const int bm = g_gemm_variant == 60 ? 256 : 192;
            a.tiles_m = (a.M + bm - 1) / bm; a.tiles_n = a.N / 256;
            const int nt = a.tiles_m * a.tiles_n, g = nt < 3 ? nt : 3;
            static unsigned tile_ctr[2] = {0, 0};                 // zero once; the kernel re-arms it (checked below)
            a.tile_ctr = g_gemm6_dynamic ? (a.tile_ctr ? a.tile_ctr : tile_ctr) : nullptr;
            if (g_gemm_variant == 60) emu::launch(dim3(g), dim3(512), [=] { gemm6_bf16_kernel<ACT, SW, 256, false>(a); });
            else if (g_gemm_variant == 61) emu::launch(dim3(g), dim3(512), [=] { gemm6_bf16_kernel<ACT, SW, 192, false>(a); });
            else emu::launch(dim3(g), dim3(512), [=] { gemm6_bf16_kernel<ACT, SW, 192, true>(a); });
            if (tile_ctr[0] || tile_ctr[1]) fprintf(stderr, "EMU: gemm6 tile counters not re-armed (%u, %u)\n", tile_ctr[0], tile_ctr[1]);
            return;
        }
        if constexpr (!SW) {
            if (g_gemm_variant == 10 && a.N % 256 == 0) {        // gemm4 on 160 x 256 tiles (the 192-row tile without group 1's third row block)
                a.tiles_m = (a.M + 159) / 160; a.tiles_n = a.N / 256;
                emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, false, -1, 160>(a); });
                return;
            }
        }
        if (g_gemm_variant == 12 && a.N % 256 == 0) {            // gemm4 on 192 x 256 tiles
            a.tiles_m = (a.M + 191) / 192; a.tiles_n = a.N / 256;
            if (!g_no_weave4) {                                 // the 192-row tiles' default since round 5 (vl2_abi.hip)
                if (a.res == nullptr) emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, true, -1, 192, true>(a); });
                else emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, false, -1, 192, true>(a); });
                return;
            }
            if (a.res == nullptr) { emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, true, -1, 192>(a); }); return; }
            emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(512), [=] { gemm4_bf16_kernel<ACT, SW, false, false, -1, 192>(a); });
            return;
        }
    }
    if constexpr (!G) {
        if (g_gemm_variant == 2) {                              // stream-K: a small persistent grid so every path is exercised
            static std::vector<float> ws;
            static std::vector<int> flags;
            const int Gsk = 12, nt = a.K / 64, total = a.tiles_m * a.tiles_n * nt;
            ws.assign((size_t)Gsk * 64 * 256, 0.f);
            flags.assign(Gsk + 1, 0);
            a.sk_ws = ws.data(); a.sk_flags = flags.data(); a.sk_per = (total + Gsk - 1) / Gsk;
            // run contributors (higher logical ids) first: the emulator executes workgroups one after the other
            std::vector<std::pair<int, unsigned>> ord;
            for (unsigned b = 0; b < (unsigned)Gsk; ++b) {
                const int q = Gsk >> 3, r = Gsk & 7, xcd = b & 7, k = b >> 3;
                ord.push_back({(xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k, b});
            }
            std::sort(ord.rbegin(), ord.rend());
            for (auto& o : ord) emu::block_order.push_back(o.second);
            emu::launch(dim3(Gsk), dim3(256), [=] { gemm_sk_bf16_kernel<ACT, SW, F32>(a); });
            if (flags[Gsk]) fprintf(stderr, "EMU: stream-K spin timeout\n");
            return;
        }
    }
    if constexpr (!SW && !F32) {
        if (g_gemm_variant == 32) {                             // small-M 64x64 form (plain and gathered)
            if (a.row_norm_out) g_need_fin = true;                // no producer-side finalize in this kernel: vl2_gemm appends the launch
            a.tiles_m = (a.M + 63) / 64; a.tiles_n = a.N / 64;
            emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(128), [=] { gemm_s_bf16_kernel<ACT, G>(a); });
            return;
        }
    }
    if (g_gemm_variant == 16) {                                 // split-K form of the 128x128 kernel (plain and gathered)
        const int nt = a.K / 64;
        const int split = nt % 4 == 0 && nt >= 8 ? 4 : nt % 3 == 0 && nt >= 6 ? 3 : nt % 2 == 0 ? 2 : 1;
        if (split > 1) {
            if (a.row_norm_out) g_need_fin = true;
            static std::vector<float> ws;
            static std::vector<int> cnt;
            ws.assign((size_t)a.tiles_m * a.tiles_n * split * 64 * 256, 0.f);
            cnt.resize(a.tiles_m * a.tiles_n, 0);                // NOT re-zeroed between launches: the kernel re-arms them
            a.sk_ws = ws.data(); a.sk_flags = cnt.data();
            emu::launch(dim3(a.tiles_m * a.tiles_n, split), dim3(256), [=] { gemm_bf16_kernel<ACT, SW, F32, G, false, true>(a); });
            return;
        }
    }
    emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(256), [=] { gemm_bf16_kernel<ACT, SW, F32, G>(a); });
}
extern "C" int32_t vl2_gemm(const vl2_gemm_desc* d, void*) {
    if (!d || d->size != sizeof(vl2_gemm_desc)) return -1;
    const int M = d->M, N = d->N, K = d->K, act = d->act;
    if (N % 128 || K % 64) return -2;
    GemmArgs a{};
    a.A = (const bf16_t*)d->A; a.W = (const bf16_t*)d->W; a.C = d->C; a.bias = d->bias; a.res = (const bf16_t*)d->res;
    a.a_idx = d->a_idx; a.M = M; a.N = N; a.K = K; a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
    a.seg_k = d->seg_k; a.out_grp = d->out_grp; a.out_grp_pad = d->out_grp_pad; a.out_row_off = d->out_row_off;
    a.res_row_mod = d->res_row_mod; a.res_row_off = d->res_row_off; a.tiles_m = (M + 127) / 128; a.tiles_n = N / 128;
    a.idx_ld = M; a.stats_out = d->stats_out; a.stats_out_np = N / 64; a.stats_in = d->stats_in; a.stats_in_np = K / 64;
    a.norm = d->norm; a.norm_eps = d->norm_eps; a.w_colsum = d->w_colsum; a.row_norm = d->row_norm;
    a.tile_ctr = (unsigned*)d->tile_ctr;
    g_weave4 = (d->flags & VL2_GEMM_WEAVE4) != 0;
    g_no_weave4 = (d->flags & VL2_GEMM_NO_WEAVE4) != 0;
    g_need_fin = d->row_norm_out && (d->flags & VL2_GEMM_NO_TICKET);
    if (d->row_norm_out) {
        if (!d->stats_out || !d->row_ticket || (d->norm_out != 1 && d->norm_out != 2)) return -1;
        if (!g_need_fin) { a.row_norm_out = d->row_norm_out; a.row_ticket = (unsigned*)d->row_ticket; a.norm_out = d->norm_out; a.norm_out_eps = d->norm_out_eps; }
    }
    struct Fin {           // runs when vl2_gemm returns, whichever branch: the appended row_norm_finalize launch of vl2_abi.hip
        const vl2_gemm_desc* d;
        ~Fin() {
            if (!g_need_fin) return;
            const vl2_gemm_desc* q = d;
            emu::launch(dim3((q->M + 31) / 32), dim3(256), [=] { row_norm_finalize_kernel(q->stats_out, q->row_norm_out, q->M, q->N / 64, q->N, q->norm_out, q->norm_out_eps); });
            g_need_fin = false;
        }
    } fin{d};
    const bool v16 = (d->variant >= 16 && d->variant <= 22) || d->variant == 26;
    g_gemm_variant = ((d->flags & VL2_GEMM_SPLITK) || d->variant == 116) ? 16 : v16 ? 0 : d->variant;     // 116: the emulator's own knob for the split-K form (tests)     // 16 = the emulator's split-K form of the 128x128 kernel
    {   // the product's variant 16 / VL2_GEMM_MFMA16 (vl2_abi.hip vl2_gemm)
        const bool ok16 = !d->a_idx && !(d->flags & 2) && d->out_grp <= 0 && d->res_row_mod <= 0 && act == 0 && !(d->stats_out && (d->flags & 1)) && N % 256 == 0;
        if (v16 && !ok16) return -3;
        const bool set16 = (d->variant == 256 || d->variant == 224 || d->variant == 192) && !(d->flags & 1);      // that tile of the 16 x 16 x 32 set on demand
        g_mfma16 = ok16 && (v16 || ((d->variant == 0 || set16) && (d->flags & VL2_GEMM_MFMA16)));
        g_mfma16_tile = g_mfma16 && set16 ? d->variant : 0;
        g_mfma16_mode = d->variant == 26 ? 9 : v16 ? d->variant - 16 : 0;
    }
    const bool sw = d->flags & 1, f32 = d->flags & 2, g = a.a_idx != nullptr;
    if (d->flags & VL2_GEMM_FP8) {          // W8A8 on the (emulated) fp8 matrix pipe: rows of K bytes seen as K / 2 16-bit elements (vl2_abi.hip)
        if (N % 256 || K % 128 || !d->row_norm || !d->col_scale || g) return -2;
        a.K = K / 2; a.lda = d->lda / 2; a.ldw = d->ldw / 2; a.norm = 1; a.row_norm = d->row_norm; a.col_scale = d->col_scale;
        a.stats_in_np = a.K / 64; a.stats_out = nullptr; a.stats_in = nullptr; a.w_colsum = nullptr;
        a.tiles_n = N / 256;
        const int v = d->variant;
        if (v == 0 || v == 4) {
            a.tiles_m = (M + 127) / 128;
            const dim3 grid(a.tiles_m * a.tiles_n);
            if (sw) emu::launch(grid, dim3(512), [=] { gemm3_fp8_kernel<0, true, false>(a); });
            else if (f32) emu::launch(grid, dim3(512), [=] { gemm3_fp8_kernel<0, false, true>(a); });
            else if (act == 3) emu::launch(grid, dim3(512), [=] { gemm3_fp8_kernel<3, false, false>(a); });
            else if (act == 0) emu::launch(grid, dim3(512), [=] { gemm3_fp8_kernel<0, false, false>(a); });
            else return -3;
        } else if (v == 8) {
            a.tiles_m = (M + 255) / 256;
            const dim3 grid(a.tiles_m * a.tiles_n);
            if (sw) emu::launch(grid, dim3(512), [=] { gemm4_fp8_kernel<0, true, false, 256>(a); });
            else if (f32) emu::launch(grid, dim3(512), [=] { gemm4_fp8_kernel<0, false, true, 256>(a); });
            else if (act == 3) emu::launch(grid, dim3(512), [=] { gemm4_fp8_kernel<3, false, false, 256>(a); });
            else if (act == 0) emu::launch(grid, dim3(512), [=] { gemm4_fp8_kernel<0, false, false, 256>(a); });
            else return -3;
        } else if (v == 12 && !f32) {
            a.tiles_m = (M + 191) / 192;
            const dim3 grid(a.tiles_m * a.tiles_n);
            if (sw) emu::launch(grid, dim3(512), [=] { gemm4_fp8_kernel<0, true, false, 192>(a); });
            else if (act == 3) emu::launch(grid, dim3(512), [=] { gemm4_fp8_kernel<3, false, false, 192>(a); });
            else if (act == 0) emu::launch(grid, dim3(512), [=] { gemm4_fp8_kernel<0, false, false, 192>(a); });
            else return -3;
        } else return -1;
        return 0;
    }
    if (d->norm && ((!d->stats_in && !d->row_norm) || (d->norm == 2 && !d->w_colsum))) return -1;
    if (d->norm && !d->row_norm && (K % 128)) return -2;          // vl2_abi.hip: the in-GEMM reduction reads the partials as 16-byte pairs
    if (d->out_grp > 0 || d->res_row_mod > 0) {
        emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(256), [=] { gemm_bf16_kernel<0, false, false, false, true>(a); });
        return 0;
    }
    if (sw) { run_gemm<0, true, false, false>(a); return 0; }
    if (g) { if (act == 3) run_gemm<3, false, false, true>(a); else if (act == 0) run_gemm<0, false, false, true>(a); else return -3; return 0; }
    if (f32) { if (act) return -3; run_gemm<0, false, true, false>(a); return 0; }
    switch (act) {
        case 0: run_gemm<0, false, false, false>(a); break;
        case 1: run_gemm<1, false, false, false>(a); break;
        case 2: run_gemm<2, false, false, false>(a); break;
        case 3: run_gemm<3, false, false, false>(a); break;
        case 5: run_gemm<5, false, false, false>(a); break;
        default: return -3;
    }
    return 0;
}
extern "C" int32_t vl2_row_norm_finalize(const float* stats, float* row_norm, int32_t rows, int32_t np, int32_t K, int32_t norm, float eps, void*) {
    emu::launch(dim3((rows + 31) / 32), dim3(256), [=] { row_norm_finalize_kernel(stats, row_norm, rows, np, K, norm, eps); });
    return 0;
}
extern "C" int32_t vl2_row_stats(const void* x, float* stats, int32_t rows, int32_t C, int32_t ldx, void*) {
    if (C % 64) return -2;
    emu::launch(dim3((rows + 3) / 4), dim3(256), [=] { row_stats_kernel((const bf16_t*)x, stats, rows, C, ldx); });
    return 0;
}
static int32_t run_norm(NormArgs a, bool rms) {
    const int nv = (a.C + 511) / 512;
    dim3 g((a.rows + 3) / 4), blk(256);
    if (rms && a.C > 2048) {
        emu::launch(dim3(a.rows), blk, [=] { norm_wide_kernel<true, 2>(a); });
    } else if (rms) {
        if (nv <= 1) emu::launch(g, blk, [=] { norm_kernel<1, true>(a); });
        else if (nv <= 2) emu::launch(g, blk, [=] { norm_kernel<2, true>(a); });
        else emu::launch(g, blk, [=] { norm_kernel<8, true>(a); });
    } else {
        if (nv <= 1) emu::launch(g, blk, [=] { norm_kernel<1, false>(a); });
        else if (nv <= 2) emu::launch(g, blk, [=] { norm_kernel<2, false>(a); });
        else if (nv <= 3) emu::launch(g, blk, [=] { norm_kernel<3, false>(a); });
        else emu::launch(g, blk, [=] { norm_kernel<8, false>(a); });
    }
    return 0;
}
extern "C" int32_t vl2_layernorm(const void* x, void* y, const float* w, const float* b, const void* res, int32_t rows,
                                 int32_t C, int32_t ldx, int32_t ldy, int32_t ldres, float eps, int32_t silu, void*) {
    return run_norm(NormArgs{(const bf16_t*)x, (bf16_t*)y, w, b, (const bf16_t*)res, rows, C, ldx, ldy, ldres, eps, silu}, false);
}
extern "C" int32_t vl2_rmsnorm(const void* x, void* y, const float* w, int32_t rows, int32_t C, int32_t ldx, int32_t ldy,
                               float eps, void*) {
    return run_norm(NormArgs{(const bf16_t*)x, (bf16_t*)y, w, nullptr, nullptr, rows, C, ldx, ldy, 0, eps, 0}, true);
}
extern "C" int32_t vl2_patchify(const void* frames, int32_t dtype, void* out, int32_t T, int32_t H, int32_t W, int32_t P,
                                int32_t G, int32_t Kp, void*) {
    dim3 g(G, T), blk(256);
    if (dtype == 0) emu::launch(g, blk, [=] { patchify_kernel<float>((const float*)frames, (bf16_t*)out, H, W, P, G, Kp); });
    else if (dtype == 1) emu::launch(g, blk, [=] { patchify_kernel<_Float16>((const _Float16*)frames, (bf16_t*)out, H, W, P, G, Kp); });
    else emu::launch(g, blk, [=] { patchify_kernel<bf16_t>((const bf16_t*)frames, (bf16_t*)out, H, W, P, G, Kp); });
    return 0;
}
extern "C" int32_t vl2_patchify_u8(const void* frames, void* out, int32_t T, int32_t H, int32_t W, int32_t P, int32_t G, int32_t Kp,
                                   float rescale, float m0, float m1, float m2, float s0, float s1, float s2, void*) {
    U8Norm n{rescale, {m0, m1, m2}, {1.0f / s0, 1.0f / s1, 1.0f / s2}};
    emu::launch(dim3(G, T), dim3(256), [=] { patchify_u8_kernel((const unsigned char*)frames, (bf16_t*)out, H, W, P, G, Kp, n); });
    return 0;
}
extern "C" int32_t vl2_fill_cls(void* x, const void* cls_pos, int32_t T, int32_t D, int32_t rows_per_frame, void*) {
    emu::launch(dim3(T), dim3(128), [=] { fill_cls_kernel((bf16_t*)x, (const bf16_t*)cls_pos, D, rows_per_frame); });
    return 0;
}
extern "C" int32_t vl2_attn_fwd(const void* q, const void* k, const void* v, void* o, int64_t q_bs, int64_t q_hs, int32_t q_rs,
                                int64_t k_bs, int64_t k_hs, int32_t k_rs, int64_t v_bs, int64_t v_hs, int32_t v_rs,
                                int64_t o_bs, int64_t o_hs, int32_t o_rs, int32_t B, int32_t H, int32_t nq, int32_t nk,
                                int32_t group, float scale, int32_t causal, int32_t causal_off, int32_t D, int32_t variant, void*) {
    const int g_attn_kv_groups = variant;
    AttnArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, q_bs, q_hs, q_rs, k_bs, k_hs, k_rs,
               v_bs, v_hs, v_rs, o_bs, o_hs, o_rs, nq, nk, group, H, B, scale * 1.4426950408889634f, causal_off};
    dim3 g((nq + 127) / 128, H, B), blk(256);
    if (causal) g = dim3(((nq + 127) / 128) * H * B, 1, 1);
    const bool cls_peel = variant == 0 && D == 64 && !causal && nq == nk && nk > 64 && (nk - 1) % 64 == 0 && group == 1;   // as the product launcher
    if (variant == 0 && D == 128 && causal && (long)((nq + 127) / 128) * H <= 352) variant = 4;   // auto, as the product launcher
    if (variant == 0 && (D == 64 || D == 128)) variant = 3;
    if (variant == 4) {                                     // two key streams per query block (NS = 2), as the product launcher
        const bool peel4 = D == 64 && !causal && nq == nk && nk > 64 && (nk - 1) % 64 == 0 && group == 1;
        const dim3 b2(512);
        if (peel4) emu::launch(dim3((nq - 1 + 127) / 128 + ((((nq - 1) & 127) == 0 || ((nq - 1) & 127) > 96) ? 1 : 0), H, B), b2, [=] { attn2_fwd_kernel<64, false, true, 2>(a); });
        else if (D == 64 && !causal) emu::launch(g, b2, [=] { attn2_fwd_kernel<64, false, false, 2>(a); });
        else if (D == 64 && causal) emu::launch(g, b2, [=] { attn2_fwd_kernel<64, true, false, 2>(a); });
        else if (D == 128 && !causal) emu::launch(g, b2, [=] { attn2_fwd_kernel<128, false, false, 2>(a); });
        else if (D == 128 && causal) emu::launch(g, b2, [=] { attn2_fwd_kernel<128, true, false, 2>(a); });
        else return -2;
        return 0;
    }
    if (variant == 3) {                                     // second structure (k_attn2.h): LDS-DMA ring + transpose reads
        if (cls_peel) emu::launch(dim3(H, B, (nq - 1 + 127) / 128 + ((((nq - 1) & 127) == 0 || ((nq - 1) & 127) > 96) ? 1 : 0)), blk, [=] { attn2_fwd_kernel<64, false, true>(a); });
        else if (D == 64 && !causal) emu::launch(g, blk, [=] { attn2_fwd_kernel<64, false>(a); });
        else if (D == 64 && causal) emu::launch(g, blk, [=] { attn2_fwd_kernel<64, true>(a); });
        else if (D == 128 && !causal) emu::launch(g, blk, [=] { attn2_fwd_kernel<128, false>(a); });
        else if (D == 128 && causal) emu::launch(g, blk, [=] { attn2_fwd_kernel<128, true>(a); });
        else return -2;
        return 0;
    }
    if (D == 64 && !causal) emu::launch(g, blk, [=] { attn_fwd_kernel<64, false>(a); });
    else if (D == 128 && causal) {
        const long per_seq = (long)((nq + 127) / 128) * H;
        if (g_attn_kv_groups == 2 || (g_attn_kv_groups == 0 && per_seq <= 352)) emu::launch(g, dim3(512), [=] { attn_fwd_kernel<128, true, 2>(a); });
        else emu::launch(g, blk, [=] { attn_fwd_kernel<128, true>(a); });
    }
    else if (D == 128 && !causal) emu::launch(g, blk, [=] { attn_fwd_kernel<128, false>(a); });
    else if (D == 64 && causal) emu::launch(g, blk, [=] { attn_fwd_kernel<64, true>(a); });
    else if (D == 96 && !causal) emu::launch(g, blk, [=] { attn_fwd_kernel<96, false>(a); });
    else return -2;
    return 0;
}
extern "C" int32_t vl2_dwconv3x3_ln_silu(const void* x, void* y, const float* wt, const float* lnw, const float* lnb, int32_t F,
                                         int32_t H, int32_t W, int32_t C, float eps, void*) {
    dim3 g(F * H * W), blk(256);
    if (W >= 16) {
        dim3 g4(F * H * ((W + DW_P - 1) / DW_P));
        if (C <= 2048) emu::launch(g4, blk, [=] { dwconv4_ln_silu_kernel<1>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
        else if (C <= 4096) emu::launch(g4, blk, [=] { dwconv4_ln_silu_kernel<2>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
        else emu::launch(g4, blk, [=] { dwconv4_ln_silu_kernel<4>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
        return 0;
    }
    if (C <= 2048) emu::launch(g, blk, [=] { dwconv_ln_silu_kernel<1>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
    else if (C <= 4096) emu::launch(g, blk, [=] { dwconv_ln_silu_kernel<2>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
    else emu::launch(g, blk, [=] { dwconv_ln_silu_kernel<4>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
    return 0;
}
#define DWS_TPF_MAX 96
static inline int dws_tpf(int H, int W) {
    const int U = H * ((W + DWS_P - 1) / DWS_P);
    return U <= DWS_TPF_MAX ? U : (U + (U + 63) / 64 - 1) / ((U + 63) / 64);
}
extern "C" int64_t vl2_dwconv_mean_workspace_bytes(int32_t F, int32_t C) { return F > 0 && C > 0 ? (int64_t)F * DWS_TPF_MAX * C * 4 : -1; }
extern "C" int32_t vl2_dwconv3x3_ln_silu_mean(const void* x, void* y, const float* wt, const float* lnw, const float* lnb, int32_t F, int32_t H,
                                              int32_t W, int32_t C, float eps, float* mean, void* ws, int64_t ws_bytes, void*) {
    const int U = H * ((W + DWS_P - 1) / DWS_P), tpf = dws_tpf(H, W), iters = (U + tpf - 1) / tpf;
    if (mean && (!ws || ws_bytes < (int64_t)F * tpf * C * 4)) return -1;
    float* psum = mean ? (float*)ws : nullptr;
    const dim3 g((F * tpf + 1) / 2), blk(512);
    if (C <= 2048) emu::launch(g, blk, [=] { dwconv_strip_ln_silu_kernel<1>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, psum, F, H, W, C, eps, tpf, iters); });
    else if (C <= 4096) emu::launch(g, blk, [=] { dwconv_strip_ln_silu_kernel<2>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, psum, F, H, W, C, eps, tpf, iters); });
    else emu::launch(g, blk, [=] { dwconv_strip_ln_silu_kernel<4>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, psum, F, H, W, C, eps, tpf, iters); });
    if (mean) emu::launch(dim3((C + 255) / 256, F), dim3(256), [=] { chan_psum_finish_kernel(psum, mean, tpf, C, 1.0f / (float)(H * W)); });
    return 0;
}
extern "C" int32_t vl2_se_excite_scale(void* x, const float* g1, const void* W2, const float* b2, int32_t F, int32_t HW, int32_t C, int32_t rd, void*) {
    if (C % 8 || rd % 16) return -2;
    emu::launch(dim3((C + 127) / 128, F), dim3(256), [=] { se_excite_scale_kernel((bf16_t*)x, g1, (const bf16_t*)W2, b2, HW, C, rd); });
    return 0;
}
extern "C" int32_t vl2_chan_mean(const void* x, float* mean, int32_t F, int32_t HW, int32_t C, void*) {
    emu::launch(dim3(C / 64, F), dim3(256), [=] { chan_mean_kernel((const bf16_t*)x, mean, HW, C); });
    return 0;
}
extern "C" int32_t vl2_small_linear(const float* x, const void* W, const float* b, float* out, int32_t F, int32_t N, int32_t K,
                                    int32_t act, void*) {
    const int a = act == 3 ? 1 : act == 4 ? 2 : 0;
    emu::launch(dim3((N + SL_NB - 1) / SL_NB, (F + 7) / 8), dim3(256), [=] { small_linear_kernel(x, (const bf16_t*)W, b, out, F, N, K, a); });
    return 0;
}
extern "C" int32_t vl2_se_scale(void* x, const float* gate, int32_t F, int32_t HW, int32_t C, void*) {
    size_t nvec = (size_t)F * HW * C / 8;
    emu::launch(dim3(7), dim3(256), [=] { se_scale_kernel((bf16_t*)x, gate, HW, C, nvec); });
    return 0;
}
extern "C" int32_t vl2_pack_fold_norm(const void* W, const void* g, const void* beta, const void* bias, void* Wp, float* colsum, float* shift,
                                      int32_t N, int32_t K, int32_t ldw, void*) {
    if ((beta != nullptr) != (shift != nullptr)) return -1;
    emu::launch(dim3((N + 3) / 4), dim3(256), [=] { pack_fold_norm_kernel((const bf16_t*)W, (const bf16_t*)g, (const bf16_t*)beta, (const bf16_t*)bias, (bf16_t*)Wp, colsum, shift, N, K, ldw); });
    return 0;
}
extern "C" int32_t vl2_pack_gate_up(const void* gate, const void* up, void* out, int32_t I, int32_t D, void*) {
    if (I % 32 || D % 8) return -2;
    emu::launch(dim3(2 * I), dim3(128), [=] { pack_gate_up_kernel((const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)out, D); });
    return 0;
}
extern "C" int32_t vl2_pack_permute(const void* in, void* out, int32_t A, int32_t B, int32_t C, int32_t out_f32, void*) {
    const dim3 grid((B * C + 255) / 256, A);
    if (out_f32) emu::launch(grid, dim3(256), [=] { pack_permute_kernel<true>((const bf16_t*)in, out, B, C); });
    else emu::launch(grid, dim3(256), [=] { pack_permute_kernel<false>((const bf16_t*)in, out, B, C); });
    return 0;
}
extern "C" int32_t vl2_pack_pad_rows(const void* in, void* out, int64_t rows, int64_t cs, int64_t cd, void*) {
    if (cd < cs) return -1;
    emu::launch(dim3((unsigned)rows), dim3(256), [=] { pack_pad_rows_kernel((const bf16_t*)in, (bf16_t*)out, (long)cs, (long)cd); });
    return 0;
}
extern "C" int32_t vl2_pack_cvt_f32(const void* in, float* out, int64_t n, void*) {
    emu::launch(dim3(3), dim3(256), [=] { pack_cvt_f32_kernel((const bf16_t*)in, out, (long)n); });
    return 0;
}
extern "C" int32_t vl2_rope_kv(const void* qkv, void* q_out, void* kc, void* vc, const float* cos_t, const float* sin_t,
                               int32_t S, int32_t nh, int32_t nkv, int32_t smax, int32_t pos0, void*) {
    if (pos0 < 0 || pos0 + S > smax) return -2;
    emu::launch(dim3(5), dim3(256), [=] { rope_kv_kernel((const bf16_t*)qkv, (bf16_t*)q_out, (bf16_t*)kc, (bf16_t*)vc, cos_t, sin_t, S, nh, nkv, smax, pos0); });
    return 0;
}
extern "C" int32_t vl2_gemv_bf16(const void* W, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                                 int32_t N, int32_t K, int32_t ldw, float eps, int32_t flags, void*) {
    GemvArgs a{(const bf16_t*)W, (const bf16_t*)x, norm_w, (const bf16_t*)res, y, N, K, ldw, eps, bias, 0, 0, 0, 0};
    if (flags & VL2_GEMV_RMS_PLAIN) { a.norm_w = nullptr; a.rms_plain = 1; }
    const bool sw = flags & 1, f32 = flags & 2;
    const int n_out = sw ? N / 2 : N;
    dim3 g((n_out + 7) / 8), blk(256);
    if (!sw && (a.norm_w || a.rms_plain) && K <= 4096) {          // vl2_abi.hip launch_gemv: the x-first form for norm-carrying single-pass rows
        dim3 g4((n_out + 3) / 4);
        if (f32) emu::launch(g4, blk, [=] { gemv_xfirst_bf16_kernel<false, true, 2>(a); });
        else emu::launch(g4, blk, [=] { gemv_xfirst_bf16_kernel<false, false, 2>(a); });
        return 0;
    }
    if (sw) emu::launch(g, blk, [=] { gemv_bf16_kernel<true, false, 2>(a); });
    else if (f32) emu::launch(g, blk, [=] { gemv_bf16_kernel<false, true, 2>(a); });
    else emu::launch(g, blk, [=] { gemv_bf16_kernel<false, false, 2>(a); });
    return 0;
}
extern "C" int32_t vl2_pack_quant_fp8(const void* w, int64_t N, int64_t K, int64_t ldw, void* q, float* scale, void*) {
    if (K % 16 || ldw % 8 || ldw < K) return -2;
    emu::launch(dim3((unsigned)N), dim3(256), [=] { quant_fp8_rows_kernel((const bf16_t*)w, (uint8_t*)q, scale, (int)K, (long)ldw); });
    return 0;
}
extern "C" int32_t vl2_quant_act_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* row_tab, int32_t M, int32_t K, int32_t norm, float eps, void*) {
    if (K % 16 || ldx % 8 || ldq % 16 || (norm != 0 && norm != 1)) return -2;
    emu::launch(dim3((unsigned)M), dim3(256), [=] { quant_act_fp8_kernel((const bf16_t*)x, (long)ldx, (uint8_t*)q, (long)ldq, row_tab, K, norm == 1 ? 1 : 0, eps); });
    return 0;
}
extern "C" int32_t vl2_gemv_fp8(const void* q, const float* scale, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                                int32_t N, int32_t K, int32_t ldq, float eps, int32_t flags, void*) {
    if (K % 16 || ldq % 16 || ldq < K || N % 2) return -2;
    Gemv8Args a{(const uint8_t*)q, scale, (const bf16_t*)x, norm_w, (const bf16_t*)res, y, N, K, ldq, eps, bias, 0};
    if (flags & VL2_GEMV_RMS_PLAIN) { a.norm_w = nullptr; a.rms_plain = 1; }
    const bool sw = flags & 1, f32 = flags & 2;
    dim3 blk(256);
#define EMU_G8(SW, F32, NP) emu::launch(dim3((N / 2 + 4 * NP - 1) / (4 * NP)), blk, [=] { gemv_fp8_kernel<SW, F32, NP>(a); })
    if (K <= 4096) { if (sw) EMU_G8(true, false, 2); else if (f32) EMU_G8(false, true, 2); else EMU_G8(false, false, 2); }
    else           { if (sw) EMU_G8(true, false, 1); else if (f32) EMU_G8(false, true, 1); else EMU_G8(false, false, 1); }
#undef EMU_G8
    return 0;
}
extern "C" int32_t vl2_gemm_skinny_bf16(const void* A, const void* W, void* C, const float* bias, const void* res, int32_t M, int32_t N,
                                        int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t ldres, int32_t flags, void* ws, int64_t ws_bytes, void*) {
    if (!ws || ws_bytes <= 0) return -1;
    const bool sw = flags & 1, f32 = flags & 2;
    if (M > 64 || N % 64 || K % 32) return -2;
    const int mt = M <= 16 ? 1 : M <= 32 ? 2 : 4, Mp = 16 * mt, steps = K / 32;
    int ks = steps % 3 == 0 ? 3 : steps % 2 == 0 ? 2 : 1;       // a small odd / even split exercises the slice + chunk loops
    const int kslice = K / ks;
    int kchunk = kslice % 64 == 0 && kslice > 64 ? kslice / 2 : kslice;   // two chunks per slice where possible
    static std::vector<float> part;
    part.assign((size_t)ks * Mp * N, 0.f);
    SkinnyArgs a{(const bf16_t*)A, (const bf16_t*)W, part.data(), M, N, K, lda, ldw, kslice, kchunk};
    if (mt == 1) emu::launch(dim3(N / 64, ks), dim3(256), [=] { gemm_skinny_kernel<1>(a); });
    else if (mt == 2) emu::launch(dim3(N / 64, ks), dim3(256), [=] { gemm_skinny_kernel<2>(a); });
    else emu::launch(dim3(N / 64, ks), dim3(256), [=] { gemm_skinny_kernel<4>(a); });
    SkinnyReduceArgs r{part.data(), C, bias, (const bf16_t*)res, M, Mp, N, ks, ldc, ldres};
    const int ncol = sw ? N / 2 : N;
    dim3 g((M * (ncol / 4) + 255) / 256), blk(256);
    if (sw) emu::launch(g, blk, [=] { skinny_reduce_kernel<true, false>(r); });
    else if (f32) emu::launch(g, blk, [=] { skinny_reduce_kernel<false, true>(r); });
    else emu::launch(g, blk, [=] { skinny_reduce_kernel<false, false>(r); });
    return 0;
}
extern "C" int32_t vl2_gemv_batched_bf16(const void* W, const void* x, const float* norm_w, const void* res, const float* bias,
                                         void* y, int32_t MB, int32_t N, int32_t K, int32_t ldw, int32_t ldx, int32_t ldy,
                                         int32_t ldres, float eps, int32_t flags, void*) {
    const bool sw = flags & 1, f32 = flags & 2;
    const int n_out = sw ? N / 2 : N, esz = f32 ? 4 : 2;
    static std::vector<unsigned char> lds;
    for (int b0 = 0; b0 < MB;) {
        const int mb = MB - b0 < 3 ? MB - b0 : 3;                // 3 rows per pass: exercises the split and the odd width
        GemvArgs a{(const bf16_t*)W, (const bf16_t*)x + (size_t)b0 * ldx, norm_w, res ? (const bf16_t*)res + (size_t)b0 * ldres : nullptr,
                   (char*)y + (size_t)b0 * ldy * esz, N, K, ldw, eps, bias, ldx, ldy, ldres};
        dim3 g((n_out + 7) / 8), blk(256);                       // 2 rows per wave
        if (mb == 1) {
            dim3 g1((n_out + 7) / 8);
            if (sw) emu::launch(g1, blk, [=] { gemv_bf16_kernel<true, false, 2>(a); });
            else if (f32) emu::launch(g1, blk, [=] { gemv_bf16_kernel<false, true, 2>(a); });
            else emu::launch(g1, blk, [=] { gemv_bf16_kernel<false, false, 2>(a); });
        } else if (mb == 2) {
            if (sw) emu::launch(g, blk, [=] { gemv_mr_bf16_kernel<true, false, 2, 2>(a); });
            else if (f32) emu::launch(g, blk, [=] { gemv_mr_bf16_kernel<false, true, 2, 2>(a); });
            else emu::launch(g, blk, [=] { gemv_mr_bf16_kernel<false, false, 2, 2>(a); });
        } else {
            if (sw) emu::launch(g, blk, [=] { gemv_mr_bf16_kernel<true, false, 3, 2>(a); });
            else if (f32) emu::launch(g, blk, [=] { gemv_mr_bf16_kernel<false, true, 3, 2>(a); });
            else emu::launch(g, blk, [=] { gemv_mr_bf16_kernel<false, false, 3, 2>(a); });
        }
        b0 += mb;
    }
    return 0;
}
extern "C" int32_t vl2_attn_decode(const void* qkv, void* kc, void* vc, const float* cos_t, const float* sin_t, float* partial,
                                   void* out, int32_t nh, int32_t nkv, int32_t smax, int32_t pos, const int32_t* pos_dev,
                                   int32_t ctx_cap, float scale, void*) {
    const int group = nh / nkv, cap = pos_dev ? ctx_cap : pos + 1, nsplit = (cap + 63) / 64;
    if (cap <= 0 || cap > smax) return -2;
    emu::launch(dim3(nsplit, nkv, (group + 3) / 4), dim3(256), [=] {
        attn_decode_kernel<false>((const bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, cos_t, sin_t, partial, nh, group, nkv, smax, pos, pos_dev, scale * 1.4426950408889634f, 0L, 0L, 0L, (int*)nullptr, (bf16_t*)nullptr); });
    emu::launch(dim3(nh), dim3(128), [=] { attn_decode_combine_kernel(partial, (bf16_t*)out, nsplit, pos, pos_dev, 0L, 0L); });
    return 0;
}
// vl2_abi.hip attn_decode_fused (attention + combine in one launch; the emulator runs the workgroups one after the other, so the
// last one in launch order of every kv head draws the last ticket)
static int32_t attn_decode_fused(const void* qkv, void* kc, void* vc, const float* cos_t, const float* sin_t, float* partial, void* out,
                                 int32_t nh, int32_t nkv, int32_t smax, const int32_t* pos_dev, float scale, int32_t* cnt, void*) {
    const int group = nh / nkv, nsplit = (smax + 63) / 64;
    emu::launch(dim3(nsplit, nkv, (group + 3) / 4), dim3(256), [=] {
        attn_decode_kernel<true>((const bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, cos_t, sin_t, partial, nh, group, nkv, smax, 0, pos_dev, scale * 1.4426950408889634f, 0L, 0L, 0L, (int*)cnt, (bf16_t*)out); });
    return 0;
}
static int32_t argmax_and_clear(const float* logits, int32_t V, int32_t* tok, int32_t* hist, int32_t* state, int32_t* zero, int32_t nzero,
                                const void* embed, void* x0, int32_t D, void*) {
    emu::launch(dim3(1), dim3(1024), [=] { argmax_kernel(logits, V, tok, hist, 0, state, (int*)zero, nzero, (const bf16_t*)embed, (bf16_t*)x0, D); });
    return 0;
}
// vl2_abi.hip vl2_decode_tail: the emulator runs workgroups one after the other, so the engine's three phases are three launches of the
// SAME kernel source (single-phase instantiations: no grid barrier), a handful of workgroups each
extern "C" int32_t vl2_decode_tail(const void* Wo, const void* Wgu, const void* Wd, int32_t ldwo, int32_t ldwgu, int32_t ldwd, const void* o,
                                   const void* x0, void* x1, void* act, void* xout, int32_t D, int32_t QD, int32_t I, float eps, int32_t* bar, void*) {
    if (!Wo || !Wgu || !Wd || !o || !x0 || !x1 || !act || !xout || !bar || D % 8 || QD % 8 || I % 32) return -1;
    TailArgs a{(const bf16_t*)Wo, (const bf16_t*)Wgu, (const bf16_t*)Wd, ldwo, ldwgu, ldwd, (const bf16_t*)o, (const bf16_t*)x0, (bf16_t*)x1,
               (bf16_t*)act, (bf16_t*)xout, D, QD, I, eps, (unsigned*)bar};
    emu::launch(dim3(8), dim3(1024), [=] { decode_tail_kernel<1>(a); });
    emu::launch(dim3(8), dim3(1024), [=] { decode_tail_kernel<2>(a); });
    emu::launch(dim3(8), dim3(1024), [=] { decode_tail_kernel<4>(a); });
    return 0;
}
extern "C" int32_t vl2_attn_decode_fused(const void* qkv, void* kc, void* vc, const float* cos_t, const float* sin_t, float* partial, void* out,
                                         int32_t nh, int32_t nkv, int32_t smax, const int32_t* pos_dev, float scale, int32_t* cnt, void* st) {
    return attn_decode_fused(qkv, kc, vc, cos_t, sin_t, partial, out, nh, nkv, smax, pos_dev, scale, cnt, st);
}

extern "C" int32_t vl2_attn_decode_batched(const void* qkv, void* kc, void* vc, const float* cos_t, const float* sin_t, float* partial,
                                           void* out, int32_t B, int64_t qkv_bs, int64_t cache_bs, int64_t out_bs, int32_t nh, int32_t nkv,
                                           int32_t smax, const int32_t* pos_dev, int32_t ctx_cap, float scale, void*) {
    const int group = nh / nkv, nsplit = (ctx_cap + 63) / 64;
    const long pbs = (long)nh * nsplit * 130;
    emu::launch(dim3(nsplit, nkv * B, (group + 3) / 4), dim3(256), [=] {
        attn_decode_kernel<false>((const bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, cos_t, sin_t, partial, nh, group, nkv, smax, 0, pos_dev, scale * 1.4426950408889634f, (long)qkv_bs, (long)cache_bs, pbs, (int*)nullptr, (bf16_t*)nullptr); });
    emu::launch(dim3(nh, B), dim3(128), [=] { attn_decode_combine_kernel(partial, (bf16_t*)out, nsplit, 0, pos_dev, pbs, (long)out_bs); });
    return 0;
}
extern "C" int32_t vl2_argmax(const float* logits, int32_t V, int32_t* tok, int32_t* hist, int32_t step, int32_t* state, void*) {
    emu::launch(dim3(1), dim3(1024), [=] { argmax_kernel(logits, V, tok, hist, step, state, (int*)nullptr, 0, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0); });
    return 0;
}
extern "C" int32_t vl2_sample_token(const float* logits, int32_t V, float temperature, int32_t top_k, float top_p, const float* u, int32_t* tok,
                                    int32_t* hist, int32_t step, int32_t* state, float* dbg, void*) {
    if (!logits || !tok || !u || V <= 0 || !(temperature > 0.f) || top_k < 0 || !(top_p > 0.f)) return -1;
    SampleArgs a{logits, V, temperature, top_k, top_p, u, tok, hist, step, state, dbg};
    if (V <= 32768) emu::launch(dim3(1), dim3(1024), [=] { sample_token_kernel<true>(a); });
    else emu::launch(dim3(1), dim3(1024), [=] { sample_token_kernel<false>(a); });
    return 0;
}
extern "C" int32_t vl2_embed_rows(const int32_t* ids, const void* table, void* out, int32_t n, int32_t D, int32_t ldo, void*) {
    emu::launch(dim3(n), dim3(128), [=] { embed_rows_kernel(ids, (const bf16_t*)table, (bf16_t*)out, D, ldo); });
    return 0;
}

#include <cstdarg>
#include <cstdio>
static int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#include "vl2_stage.inc"

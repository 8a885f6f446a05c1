// TEST INFRASTRUCTURE ONLY: compiles the product's kernel headers against tests/emu/hip_emu.h (host CPU) so
// tests/test_emu_kernels.py can check their indexing logic without a GPU.  Built by tests/emu/build_emu.py.
#include "hip_emu.h"
alignas(64) unsigned char vl2_smem[160 * 1024];
#include "k_gemm.h"
#include "k_norm.h"
#include "k_vit.h"
#include "k_attn.h"
#include "k_stc.h"
#include "k_decode.h"

template <int ACT, bool SW, bool F32, bool G>
static void run_gemm(GemmArgs a) {
    emu::launch(dim3(a.tiles_m * a.tiles_n), dim3(256), [=] { gemm_bf16_kernel<ACT, SW, F32, G>(a); });
}
extern "C" int emu_gemm(const void* A, const void* W, void* C, const float* bias, const void* res, const int* a_idx,
                        const void* zero_row, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int seg_k,
                        int out_grp, int out_grp_pad, int out_row_off, int res_row_mod, int res_row_off, int act,
                        int swiglu, int out_f32) {
    GemmArgs a{(const bf16_t*)A, (const bf16_t*)W, C, bias, (const bf16_t*)res, a_idx, (const bf16_t*)zero_row,
               M, N, K, lda, ldw, ldc, ldres, seg_k, out_grp, out_grp_pad, out_row_off, res_row_mod, res_row_off,
               (M + 127) / 128, N / 128};
    const bool g = a_idx != nullptr;
    if (swiglu) { run_gemm<0, true, false, false>(a); return 0; }
    if (g) { if (act == 3) run_gemm<3, false, false, true>(a); else run_gemm<0, false, false, true>(a); return 0; }
    if (out_f32) { run_gemm<0, false, true, false>(a); return 0; }
    switch (act) {
        case 0: run_gemm<0, false, false, false>(a); break;
        case 1: run_gemm<1, false, false, false>(a); break;
        case 2: run_gemm<2, false, false, false>(a); break;
        case 3: run_gemm<3, false, false, false>(a); break;
        default: return -1;
    }
    return 0;
}

extern "C" int emu_norm(const void* x, void* y, const float* w, const float* b, const void* res, int rows, int C, int ldx,
                        int ldy, int ldres, float eps, int silu, int rms) {
    NormArgs a{(const bf16_t*)x, (bf16_t*)y, w, b, (const bf16_t*)res, rows, C, ldx, ldy, ldres, eps, silu};
    const int nv = (C + 511) / 512;
    dim3 g((rows + 3) / 4), blk(256);
    if (rms) {
        if (nv <= 1) emu::launch(g, blk, [=] { norm_kernel<1, true>(a); });
        else if (nv <= 2) emu::launch(g, blk, [=] { norm_kernel<2, true>(a); });
        else emu::launch(g, blk, [=] { norm_kernel<8, true>(a); });
    } else {
        if (nv <= 1) emu::launch(g, blk, [=] { norm_kernel<1, false>(a); });
        else if (nv <= 2) emu::launch(g, blk, [=] { norm_kernel<2, false>(a); });
        else emu::launch(g, blk, [=] { norm_kernel<8, false>(a); });
    }
    return 0;
}
extern "C" int emu_patchify(const void* frames, int dtype, void* out, int T, int H, int W, int P, int G, int Kp) {
    dim3 g(G, T), blk(256);
    if (dtype == 0) emu::launch(g, blk, [=] { patchify_kernel<float>((const float*)frames, (bf16_t*)out, H, W, P, G, Kp); });
    else if (dtype == 1) emu::launch(g, blk, [=] { patchify_kernel<_Float16>((const _Float16*)frames, (bf16_t*)out, H, W, P, G, Kp); });
    else emu::launch(g, blk, [=] { patchify_kernel<bf16_t>((const bf16_t*)frames, (bf16_t*)out, H, W, P, G, Kp); });
    return 0;
}
extern "C" int emu_attn(const void* q, const void* k, const void* v, void* o, long q_bs, long q_hs, int q_rs, long k_bs,
                        long k_hs, int k_rs, long v_bs, long v_hs, int v_rs, long o_bs, long o_hs, int o_rs, int B, int H,
                        int nq, int nk, int group, float scale, int causal, int causal_off, int D) {
    AttnArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, q_bs, q_hs, q_rs, k_bs, k_hs, k_rs,
               v_bs, v_hs, v_rs, o_bs, o_hs, o_rs, nq, nk, group, scale * 1.4426950408889634f, causal_off};
    dim3 g((nq + 127) / 128, H, B), blk(256);
    if (D == 64 && !causal) emu::launch(g, blk, [=] { attn_fwd_kernel<64, false>(a); });
    else if (D == 128 && causal) emu::launch(g, blk, [=] { attn_fwd_kernel<128, true>(a); });
    else if (D == 128 && !causal) emu::launch(g, blk, [=] { attn_fwd_kernel<128, false>(a); });
    else if (D == 64 && causal) emu::launch(g, blk, [=] { attn_fwd_kernel<64, true>(a); });
    else return -1;
    return 0;
}

extern "C" int emu_dwconv_ln_silu(const void* x, void* y, const float* wt, const float* lnw, const float* lnb, int F, int H,
                                  int W, int C, float eps) {
    dim3 g(F * H * W), blk(256);
    if (C <= 2048) emu::launch(g, blk, [=] { dwconv_ln_silu_kernel<1>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
    else emu::launch(g, blk, [=] { dwconv_ln_silu_kernel<2>((const bf16_t*)x, (bf16_t*)y, wt, lnw, lnb, H, W, C, eps); });
    return 0;
}
extern "C" int emu_chan_mean(const void* x, float* mean, int F, int HW, int C) {
    emu::launch(dim3(C / 64, F), dim3(256), [=] { chan_mean_kernel((const bf16_t*)x, mean, HW, C); });
    return 0;
}
extern "C" int emu_small_linear(const float* x, const void* W, const float* b, float* out, int F, int N, int K, int act) {
    emu::launch(dim3((N + 3) / 4), dim3(256), [=] { small_linear_kernel(x, (const bf16_t*)W, b, out, F, N, K, act); });
    return 0;
}
extern "C" int emu_se_scale(void* x, const float* gate, int F, int HW, int C) {
    size_t nvec = (size_t)F * HW * C / 8;
    emu::launch(dim3(7), dim3(256), [=] { se_scale_kernel((bf16_t*)x, gate, HW, C, nvec); });
    return 0;
}
extern "C" int emu_rope_kv(const void* qkv, void* q_out, void* kc, void* vc, const float* cos_t, const float* sin_t, int S,
                           int nh, int nkv, int smax, int pos0) {
    emu::launch(dim3(5), dim3(256), [=] { rope_kv_kernel((const bf16_t*)qkv, (bf16_t*)q_out, (bf16_t*)kc, (bf16_t*)vc, cos_t, sin_t, S, nh, nkv, smax, pos0); });
    return 0;
}
extern "C" int emu_gemv(const void* W, const void* x, const float* norm_w, const void* res, void* y, int N, int K, int ldw,
                        float eps, int swiglu, int out_f32) {
    GemvArgs a{(const bf16_t*)W, (const bf16_t*)x, norm_w, (const bf16_t*)res, y, N, K, ldw, eps};
    const int n_out = swiglu ? N / 2 : N;
    dim3 g((n_out + 7) / 8), blk(256);
    if (swiglu) emu::launch(g, blk, [=] { gemv_bf16_kernel<true, false, 2>(a); });
    else if (out_f32) emu::launch(g, blk, [=] { gemv_bf16_kernel<false, true, 2>(a); });
    else emu::launch(g, blk, [=] { gemv_bf16_kernel<false, false, 2>(a); });
    return 0;
}
extern "C" int emu_attn_decode(const void* q, const void* kc, const void* vc, float* partial, void* out, int nh, int nkv,
                               int smax, int ctx, int chunk, float scale) {
    const int group = nh / nkv, nsplit = (ctx + chunk - 1) / chunk;
    emu::launch(dim3(nsplit, nkv), dim3(group * 64), [=] {
        attn_decode_kernel((const bf16_t*)q, (const bf16_t*)kc, (const bf16_t*)vc, partial, nh, group, smax, ctx, chunk, scale * 1.4426950408889634f); });
    emu::launch(dim3(nh), dim3(128), [=] { attn_decode_combine_kernel(partial, (bf16_t*)out, nsplit); });
    return 0;
}
extern "C" int emu_argmax(const float* logits, int V, int* tok, int* hist, int step) {
    emu::launch(dim3(1), dim3(1024), [=] { argmax_kernel(logits, V, tok, hist, step); });
    return 0;
}
extern "C" int emu_embed_rows(const int* ids, const void* table, void* out, int n, int D, int ldo) {
    emu::launch(dim3(n), dim3(128), [=] { embed_rows_kernel(ids, (const bf16_t*)table, (bf16_t*)out, D, ldo); });
    return 0;
}

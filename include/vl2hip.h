/* libvl2hip.so -- C ABI of the MI355X-native VideoLLaMA2 video-inference hot path.
 *
 * The reference (DAMO-NLP-SG/VideoLLaMA2) is pure Python and has NO FFI of its own: every FLOP of its hot path runs
 * inside third-party library kernels (HF transformers / timm / torch / flash-attn).  Each entry
 * point below therefore cites the reference call site (file:line under /root/reference, or HF: for the installed
 * transformers package) whose delegated library kernel it replaces.  The Python host code in videollama2_amd/ binds
 * these with ctypes (videollama2_amd/_lib.py); INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int32: 0 = ok, < 0 = VL2_E_* (argument/shape error, nothing launched), > 0 = hipError_t.
 *   - no allocation, no synchronisation, no host<->device copies inside: every call only enqueues kernels on `stream`
 *     (a hipStream_t passed as void*), so every call is hipGraph-capturable.  The caller owns all buffers.
 *   - all pointers are DEVICE pointers unless stated; bf16 tensors are raw uint16 bit patterns; strides (ld*) are in
 *     elements; every row pointer / ld must keep 16-byte alignment (8 bf16).
 *   - thread-safe and re-entrant: the library holds NO mutable process state.  The only state is thread-local (the error
 *     string below); kernel variants, split-K and workspaces are per-call arguments.
 */
#ifndef VL2HIP_H
#define VL2HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VL2_ABI_VERSION 7   /* 7 (round 6): the lab entry points left the product library; VL2_STAGE_NO_MFMA16; gate/up of vl2_llm_prefill on the 16x16x32 kernels by default */
#define VL2_E_BADARG  (-1)   /* null pointer / non-positive size */
#define VL2_E_SHAPE   (-2)   /* shape not supported by the gfx950 kernels (alignment / multiple-of constraints) */
#define VL2_E_UNSUPP  (-3)   /* option combination not built */

int32_t vl2_version(void);
/* The 16-bit element type this build of the library computes in: "bf16" (libvl2hip.so) or "fp16" (libvl2hip_f16.so, the same sources with
 * -DVL2_ELEM_F16).  Every `*_bf16` entry point, every 16-bit buffer and weight of a descriptor is in THIS type; fp32 / int arguments do not
 * change.  (The reference picks the type per checkpoint load: /root/reference/videollama2/model/__init__.py:71 torch_dtype=float16.) */
const char* vl2_elem_name(void);
const char* vl2_last_error_string(void);      /* host pointer, thread-local, valid until the next failing call */
/* Size of the caller-owned device workspace that `vl2_gemm` (split-K / stream-K forms) and `vl2_gemm_skinny_bf16` take per
 * call.  It must be zero-filled once when allocated (split-K tile counters live in it and are re-armed by the kernels) and
 * calls sharing one workspace must be ordered on one stream; use one workspace per stream for concurrent streams. */
int64_t vl2_workspace_bytes(void);

/* activation codes for vl2_gemm / vl2_small_linear */
#define VL2_ACT_NONE    0
#define VL2_ACT_QGELU   1   /* x*sigmoid(1.702x)  HF:activations.py QuickGELUActivation (CLIP MLP) */
#define VL2_ACT_GELU    2   /* exact erf GELU     torch nn.GELU() in projector.py:125-130 build_mlp (STC readout) */
#define VL2_ACT_SILU    3   /* nn.SiLU            projector.py:164-174 sampler, timm act_layer */
#define VL2_ACT_SIGMOID 4   /* only vl2_small_linear */
#define VL2_ACT_GELU_TANH 5 /* nn.GELU(approximate='tanh') = HF gelu_pytorch_tanh (SigLIP MLP, HF:models/siglip/modeling_siglip.py SiglipMLP) */
/* flags for vl2_gemm */
#define VL2_GEMM_SWIGLU  1  /* W = blocks of 64 rows {32 gate rows, 32 up rows}; C[m, j] = silu(gate_j) * up_j; C has N/2 cols.
                               HF:models/mistral/modeling_mistral.py MistralMLP.forward */
#define VL2_GEMM_OUT_F32 2  /* C is fp32 (validation / logits) */
#define VL2_GEMM_SPLITK  4  /* small grids may split K through the workspace (`ws`): trades "a row's bits do not depend on M"
                               for latency on few-tile shapes (per-rank shapes of the frame-sharded encoder); off by default */
#define VL2_GEMM_PERSISTENT 8  /* the automatic kernel choice may take the persistent form (one workgroup per CU walks its tiles through `tile_ctr`,
                                  csrc/k_gemm6.h: bf16 output without residual / statistics / gather, K >= 1024, more than one round of 256-row tiles).
                                  vl2_vit_forward sets it for the tower's GEMMs (q/k/v, fc1: -0.15 ... -0.27 ms per 16-frame pass on 7 of 7 boxes,
                                  profiles/r05_experiments.md section 5); elsewhere off (neutral or slower in the power-limited pipeline) */
#define VL2_GEMM_FP8     128  /* W8A8 on the fp8 matrix pipe (v_mfma_f32_32x32x64_f8f6f4, twice the 16-bit MFMA rate; SURVEY 8f row 5 / BASELINE.json configs[4]
                                 "fp8 MFMA on CDNA4"): A [M, lda] and W [N, ldw] hold OCP e4m3fn BYTES (K = elements = bytes, K % 128 == 0, N % 256 == 0),
                                 `row_norm` [M][2] = (0, row multiplier) as vl2_quant_act_fp8 writes it (activation scale x RMS rstd), `col_scale` [N] = the
                                 weight rows' scales (vl2_pack_quant_fp8).  C = epilogue(rowmul_m * colscale_n * sum_k A8 W8): bias / SiLU / SwiGLU / residual /
                                 fp32 output as the 16-bit form.  An OPTIONAL arithmetic (both operands rounded to e4m3fn), never the default */
#define VL2_GEMM_MFMA16  2048 /* opt-in: plain bf16-output calls without activation (incl. SWIGLU; N % 256 == 0, no gather / remap / stats_out) run on the 256x256 ping-pong
                               * kernel built on v_mfma_f32_16x16x32_bf16 (csrc/k_gemm9.h; variant 16 = the same on demand).  The instruction sustains ~15 % more at this
                               * part's power limit than the library's v_mfma_f32_32x32x16_bf16, but sums 32 products per accumulation step instead of 16: results are as
                               * accurate but NOT bit-identical with every other variant, so a call site must use it for ALL its calls or none (a row's bits still do
                               * not depend on M).  Ignored where the kernel is not built. */
/* kernel variants (vl2_gemm_desc.variant; 0 = per-shape choice; every variant EXCEPT 16 produces the same bits): 1 = 128x128x64 two-barrier kernel,
 * 4 = 128x256x64 ping-pong, 8 = 256x256x32 ping-pong, 12 = the same on 192x256 tiles (4, 8, 12: N%256==0; 12: bf16 output), 32 = 64x64 small-M kernel,
 * 256 = 128x128 8-wave deep-ring one-round kernel, 224 / 192 = the fill-the-round 224x128 / 192x128 ping-pong kernel (csrc/k_gemm7.h; N % 128 == 0,
 * plain or gathered A, no SwiGLU), 60 / 61 = the persistent 256x256 / 192x256 kernel with a static tile walk, 70 / 71 = with tiles handed out through
 * `tile_ctr` (what VL2_GEMM_PERSISTENT selects), 24 = the automatic choice without the persistent form, 16 / 26 = the 256x256 tile on
 * v_mfma_f32_16x16x32_bf16 with 32- / 64-deep phases (see VL2_GEMM_MFMA16; a row-split call = ONE mixed launch with 128x128 tail tiles of the same instruction).  A call that does not qualify for a forced variant gets the automatic choice. */
/* stage-level descriptors (vl2_vit_desc / vl2_stc_desc / vl2_llm_desc) `flags` */
#define VL2_STAGE_PERSISTENT_GEMM  1   /* every GEMM of the stage with VL2_GEMM_PERSISTENT (vl2_vit_forward: the default, see VL2_STAGE_VIT_NO_PERSISTENT) */
#define VL2_STAGE_MFMA16       32768   /* (rounds 5: opt-in; since round 6 the default of vl2_llm_prefill -- accepted and ignored) */
#define VL2_STAGE_NO_MFMA16    65536   /* vl2_llm_prefill: the gate/up projection (the step's dominant GEMM) WITHOUT VL2_GEMM_MFMA16.  Default since round 6: gate/up runs
                                         * on v_mfma_f32_16x16x32_bf16 at every S -- ONE launch of 256 x 256 tiles + 128 x 128 tail tiles (csrc/k_gemm9.h gemm_mix16_bf16_kernel):
                                         * 354 -> 333 us at S = 1621, prefill -0.6 ms (profiles/r06_experiments.md).  The call site keeps one arithmetic for all its rows and
                                         * all M (what batched == sequential pins); its dot products associate in steps of 32 products where the rest of the family's take 16. */
#define VL2_STAGE_PREFILL_FP8     512   /* vl2_llm_prefill: the four projections of every layer as VL2_GEMM_FP8 calls on the fp8 weight copies (layers_w8), their
                                         * inputs quantised per row by vl2_quant_act_fp8 (which also computes the RMS rstd: no statistics launches).  Attention,
                                         * RoPE, the KV cache and lm_head stay 16-bit.  OPTIONAL arithmetic (W8A8), never the default, never the headline */
#define VL2_STAGE_DECODE_FP8       64   /* decode step: the five projections of every layer and lm_head read the fp8 copies of the weights
                                         * (vl2_llm_desc.layers_w8 / lm_head_w8: vl2_gemv_fp8 instead of vl2_gemv_bf16).  A different arithmetic
                                         * (weights rounded to e4m3fn): OPTIONAL, never the default, never the headline number */
#define VL2_GEMV_RMS_PLAIN 32  /* vl2_gemv_bf16 `flags`: RMS-normalise x with NO weight vector (the norm weight is folded into W; `norm_w` is ignored):
                                 the same bits as a vector of ones, without every workgroup reading 4 K bytes of ones */
#ifdef VL2_EXPERIMENTAL
/* ---- A/B SWITCHES AND LAB FORMS (not permanent ABI; a host sees them only with -DVL2_EXPERIMENTAL).  Every one of them selects another schedule of the
 * SAME arithmetic (same bits) that was measured against the default and did not win; the numbers are in profiles/r0*_experiments.md.  "lab" = compiled only
 * into libvl2hip_lab.so (-DVL2_LAB, scripts/build_lab_lib.sh): the product library ignores the flag or refuses the variant (VL2_E_UNSUPP). */
#define VL2_GEMM_NO_MIX  16   /* a row-split call stays two launches instead of ONE mixed launch (gemm_mix_bf16_kernel) */
#define VL2_GEMM_NO_FILL 32   /* the automatic choice does not take the fill-the-round kernel (variants 224 / 192) */
#define VL2_GEMM_WEAVE   64   /* lab: the 128x256 / 224x128 / 192x128 ping-pong kernels issue their LDS-DMA woven between the MFMAs of the matrix phases */
#define VL2_GEMM_NO_TICKET 256 /* `row_norm_out` is filled by a separate vl2_row_norm_finalize launch behind the GEMM instead of by the GEMM's last column tile */
#define VL2_GEMM_WEAVE4  512  /* lab: the 256x256 ping-pong kernel (and the big tiles of the mixed launch) with the woven issue the 192x256 tiles take by default */
#define VL2_GEMM_NO_WEAVE4 1024 /* lab: the 192x256 tiles WITHOUT the woven issue */
/* lab variants: 2 = stream-K form of variant 1 (needs `ws`; other summation order), 5 = variant 4 with the woven issue, 225 / 193 = 224 / 192 with it,
 * 10 = the 256-wide ping-pong kernel on 160-row tiles (round 6: fills the one-round grids of the tower's N = 1024 GEMMs better and is not faster -- the part is power-limited),
 * 9 = the 256x256 tile on FOUR waves (csrc/k_gemm8.h), 62 = 61 with two accumulator sets, 17 ... 22 = issue orders of variant 16, 23 / 25 = variant 16
 * with s_memtime stamps (scripts/gemm9_phase_stamps.py). */
#define VL2_STAGE_VIT_NO_PERSISTENT 4096 /* vl2_vit_forward only: its GEMMs WITHOUT VL2_GEMM_PERSISTENT */
#define VL2_STAGE_NO_MIX           2   /* every GEMM of the stage with VL2_GEMM_NO_MIX */
#define VL2_STAGE_NO_FILL_TILES  128   /* ... with VL2_GEMM_NO_FILL */
#define VL2_STAGE_WEAVE          256   /* ... with VL2_GEMM_WEAVE (lab) */
#define VL2_STAGE_WEAVE4        8192   /* ... with VL2_GEMM_WEAVE4 (lab) */
#define VL2_STAGE_NO_WEAVE4    16384   /* ... with VL2_GEMM_NO_WEAVE4 (lab) */
#define VL2_STAGE_ROW_TICKET    1024   /* ViT and LLM prefill: the statistics-producing GEMMs (out_proj / fc2, o / down) finalize their own output rows
                                         * (vl2_gemm_desc.row_norm_out: producer-side ticket) instead of a vl2_row_norm_finalize launch behind each of them */
#define VL2_STAGE_SELF_REDUCE      4   /* ViT and LLM prefill: the norm-carrying GEMMs reduce the row statistics themselves (no row_norm_finalize launches) */
#define VL2_STAGE_FUSED_DECODE_ATTN 8  /* lab: decode step: attention + combine as one launch (vl2_attn_decode_fused) */
#define VL2_STAGE_STC_UNFUSED      32   /* connector: the SE block as the five launches of rounds 1-3 (dwconv, chan_mean, 2 x small_linear, se_scale) instead of
                                         * vl2_dwconv3x3_ln_silu_mean + small_linear + vl2_se_excite_scale */
#define VL2_STAGE_DECODE_TAIL      16   /* lab: decode step: o_proj / gate-up / down as ONE vl2_decode_tail launch instead of three vl2_gemv_bf16 launches */
#endif /* VL2_EXPERIMENTAL (flags) */
#define VL2_NORM_NONE 0
#define VL2_NORM_RMS  1     /* HF:modeling_mistral.py MistralRMSNorm in front of q/k/v and gate/up */
#define VL2_NORM_LN   2     /* HF:modeling_clip.py layer_norm1 / layer_norm2 in front of q/k/v and fc1 */

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T):  bf16 MFMA GEMM, fp32 accumulate.  Replaces the cuBLAS / cuDNN GEMMs behind
 *   nn.Linear in HF:models/clip/modeling_clip.py CLIPAttention/CLIPMLP, HF:models/mistral/modeling_mistral.py
 *   MistralAttention/MistralMLP/lm_head, and the 1x1 convs / Conv3d / readout of videollama2/model/projector.py:153-187.
 * N % 128 == 0, K % 64 == 0.  bias fp32 [N] or NULL.  res bf16 rows or NULL (added after the activation).  Operands of any
 *   size: rows beyond the kernels' 32-bit buffer offsets (>= 2 GiB of A or W) are covered by splitting the call into row /
 *   column chunks (e.g. the 2.49 GB lm_head of VideoLLaMA2-72B); the gathered row pool itself must stay below 2 GiB.
 * Gathered-A form (a_idx != NULL): K = nseg*seg_k; the A row of (segment s, output row m) is A[a_idx[s*M+m]] or
 *   zeros when the index is < 0 (an out-of-range buffer offset, which gfx950 reads as zeros) -- this is
 *   Conv3d(k=2,s=2,p=1) as a GEMM over the 8 taps (projector.py:164-174).
 * Row remap: out_grp > 0 -> output row = m + (m/out_grp)*out_grp_pad + out_row_off; res_row_mod > 0 -> residual row =
 *   m % res_row_mod + res_row_off (else the output row).  Used by the patch-embed GEMM to write torch.cat([cls, patches])
 *   + position_embedding directly (HF:modeling_clip.py CLIPVisionEmbeddings.forward).
 * Norm-carrying chain (csrc/k_gemm.h): `stats_out` != NULL makes the epilogue emit, per output row and 64-column block,
 *   (sum, sum of squares) of the row as stored -> fp32 [M][N/64][2]; a consumer passes that buffer as `stats_in` (it then
 *   holds K/64 blocks per A row) with norm = VL2_NORM_RMS / VL2_NORM_LN and computes Norm(A) W^T on the RAW A rows:
 *   y = rstd*(A W'^T) (RMS) or rstd*(A W'^T - mean*w_colsum) + bias (LN), W' = W*g folded at pack time, bias = W b + c. */
typedef struct vl2_gemm_desc {
    uint32_t size;                 /* sizeof(vl2_gemm_desc) -- lets the struct grow without breaking callers */
    int32_t M, N, K;
    const void* A;  int32_t lda;
    const void* W;  int32_t ldw;
    void* C;        int32_t ldc;
    const float* bias;
    const void* res; int32_t ldres;
    int32_t act, flags;
    const int32_t* a_idx; int32_t seg_k;
    int32_t out_grp, out_grp_pad, out_row_off, res_row_mod, res_row_off;
    float* stats_out;
    const float* stats_in;
    int32_t norm;  float norm_eps;
    const float* w_colsum;
    const float* row_norm;         /* optional [M][2] (mean, rstd) from vl2_row_norm_finalize: replaces the per-tile reduction of stats_in */
    void* ws;  int64_t ws_bytes;   /* optional workspace (vl2_workspace_bytes()); NULL = no split-K / stream-K */
    int32_t variant;               /* 0 = auto */
    const float* col_scale;        /* VL2_GEMM_FP8 only: [N] multipliers of the output columns (else NULL) */
    void* tile_ctr;                /* optional: 8 zeroed bytes (two uint32) through which the persistent GEMM hands out its tiles; the kernel
                                      re-arms them, so one block serves every GEMM of a stream (vl2_fill_zero once).  NULL: the block at the
                                      end of `ws` if that is given, else the persistent form is used on request only (static tile walk) */
    float* row_norm_out;           /* optional, with stats_out: [M][2] (mean, rstd) of the OUTPUT rows for the GEMM that consumes them (its `row_norm`), written by
                                      this call -- by the workgroup that stores the last column tile of a row block (csrc/k_gemm.h gemm_rows_ticket: the bits of
                                      vl2_row_norm_finalize, no launch between producer and consumer), or by an appended vl2_row_norm_finalize launch where the
                                      chosen kernel has no such epilogue (64 x 64 tiles, split-K, column chunks) or VL2_GEMM_NO_TICKET is set */
    uint32_t* row_ticket;          /* with row_norm_out: M / 64 + 2 zeroed words (vl2_fill_zero once; the kernels re-arm them, so one block serves a stream) */
    int32_t norm_out;  float norm_out_eps;   /* with row_norm_out: VL2_NORM_RMS / VL2_NORM_LN of the output rows and its epsilon */
} vl2_gemm_desc;
int32_t vl2_gemm(const vl2_gemm_desc* d, void* stream);
/* (sum, sum of squares) per row and 64-column block of a bf16 activation x [rows, C] -> stats fp32 [rows][C/64][2], in the
 * layout / summation order of vl2_gemm's `stats_out`: seeds a norm-carrying chain whose first tensor no GEMM wrote
 * (inputs_embeds; the CLIP embeddings after pre_layrnorm) or re-derives it after a tensor-parallel all-reduce.  C % 64 == 0. */
int32_t vl2_row_stats(const void* x, float* stats, int32_t rows, int32_t C, int32_t ldx, void* stream);
/* p[0, bytes) = 0 (bytes % 4 == 0) as a KERNEL on `stream`: the counter blocks this library's kernels re-arm themselves (vl2_gemm_desc.tile_ctr,
 * the split-K tile counters) must start from zero, and a captured hipMemsetAsync node of this size did not replay on ROCm 7.2. */
int32_t vl2_fill_zero(void* p, int64_t bytes, void* stream);
/* stats [rows][np][2] (as vl2_gemm's stats_out / vl2_row_stats write them, np = K/64) -> row_norm [rows][2] = (mean, rstd):
 * norm = VL2_NORM_RMS (mean 0, rstd = rsqrt(sum x^2 / K + eps)) or VL2_NORM_LN.  Pass the result as vl2_gemm_desc.row_norm so
 * the consuming GEMM does not repeat the reduction in each of its column tiles. */
int32_t vl2_row_norm_finalize(const float* stats, float* row_norm, int32_t rows, int32_t np, int32_t K, int32_t norm, float eps, void* stream);

/* y = LayerNorm(x)*w + b [+ res] [-> SiLU]; rows x C, fp32 statistics.  HF:modeling_clip.py pre_layrnorm / layer_norm1/2;
 * timm LayerNormAct2d in channels-last form (projector.py:153-184) incl. the Bottleneck tail act3(conv3(x)+shortcut).
 * C % 8 == 0, C <= 4096. */
int32_t vl2_layernorm(const void* x, void* y, const float* w, const float* b, const void* res, int32_t rows, int32_t C,
                      int32_t ldx, int32_t ldy, int32_t ldres, float eps, int32_t silu, void* stream);
/* HF:modeling_mistral.py MistralRMSNorm.forward. */
int32_t vl2_rmsnorm(const void* x, void* y, const float* w, int32_t rows, int32_t C, int32_t ldx, int32_t ldy, float eps,
                    void* stream);

/* frames [T,3,H,W] (dtype 0=fp32, 1=fp16, 2=bf16; what process_video returns / mm_infer casts, videollama2/__init__.py:60)
 * -> im2col rows [T*G*G, Kp] bf16 (k = c*P*P+ky*P+kx, zero padded to Kp).  HF:modeling_clip.py CLIPVisionEmbeddings
 * patch_embedding (Conv2d k=P s=P, no bias). */
int32_t vl2_patchify(const void* frames, int32_t dtype, void* out, int32_t T, int32_t H, int32_t W, int32_t P, int32_t G,
                     int32_t Kp, void* stream);
/* uint8 ingest: frames [T,H,W,3] uint8 (device) -> the same im2col rows, with the image processor's x*rescale, (x-mean)/std
 * (videollama2/mm_utils.py:196-201 -> HF CLIPImageProcessor / SiglipImageProcessor preprocess) done in registers.
 * The per-channel mean / std travel by value. */
int32_t vl2_patchify_u8(const void* frames_thwc, void* out, int32_t T, int32_t H, int32_t W, int32_t P, int32_t G, int32_t Kp,
                        float rescale, float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream);
/* x[t*rows_per_frame, :] = cls_pos (class_embedding + position_embedding[0]). */
int32_t vl2_fill_cls(void* x, const void* cls_pos, int32_t T, int32_t D, int32_t rows_per_frame, void* stream);

/* `variant`: 0 = auto (head_dim 64 / 128: 3, or 4 for causal head_dim 128 while one sequence has <= 352 (query block, head) pairs; the
 * register-staged kernel for head_dim 96); 1 = register-staged K/V (k_attn.h), one
 * group of 4 waves per workgroup; 2 = the same with two groups that split the KV tiles and merge through LDS (causal D=128 only);
 * 3 = K/V by LDS-DMA into a two-stage ring, V through the transpose read (k_attn2.h); 4 = 3 with two key streams per query block
 * (even / odd 64-key tiles on two groups of 4 waves, states merged in fp32 through LDS; 512 threads, four LDS stages); 5 = (libvl2hip_lab.so only) 3 for
 * causal head_dim 128 without the return from tiles the mask hides from a whole wave, for A/B runs.
 * Fused attention forward, softmax in fp32.  D = 64 or 128.  Element strides: *_bs batch, *_hs head, *_rs row.
 * kv head of q head h = h / group.  causal: key j visible to q row i iff j <= i + causal_off.
 * Replaces flash-attn (videollama2/model/encoder.py:24) / HF eager_attention_forward for CLIP, and HF Mistral
 * attention (repeat_kv + causal softmax) for the prefill. */
int32_t vl2_attn_fwd(const void* q, const void* k, const void* v, void* o, int64_t q_bs, int64_t q_hs, int32_t q_rs,
                     int64_t k_bs, int64_t k_hs, int32_t k_rs, int64_t v_bs, int64_t v_hs, int32_t v_rs, int64_t o_bs,
                     int64_t o_hs, int32_t o_rs, int32_t B, int32_t H, int32_t nq, int32_t nk, int32_t group, float scale,
                     int32_t causal, int32_t causal_off, int32_t D, int32_t variant, void* stream);

/* STC connector direct kernels (channels-last activations [F, H, W, C]); timm Bottleneck.conv2 + LayerNormAct2d, SEModule. */
int32_t vl2_dwconv3x3_ln_silu(const void* x, void* y, const float* w9c, const float* lnw, const float* lnb, int32_t F,
                              int32_t H, int32_t W, int32_t C, float eps, void* stream);
int32_t vl2_chan_mean(const void* x, float* mean, int32_t F, int32_t HW, int32_t C, void* stream);
int32_t vl2_small_linear(const float* x, const void* W, const float* b, float* out, int32_t F, int32_t N, int32_t K,
                         int32_t act, void* stream);
int32_t vl2_se_scale(void* x, const float* gate, int32_t F, int32_t HW, int32_t C, void* stream);
/* The same depthwise conv + LayerNorm2d + SiLU with the SE squeeze folded in: also writes mean[F, C] = y.mean over (H, W) (timm SEModule,
 * x.mean((2, 3))), from per-team partial sums added in a fixed order (a frame's result does not depend on F).  The taps `w9c` are held in the
 * build's element type inside the kernel (exact for weights that came from a checkpoint of that type).  ws >= vl2_dwconv_mean_workspace_bytes. */
int64_t vl2_dwconv_mean_workspace_bytes(int32_t F, int32_t C);
int32_t vl2_dwconv3x3_ln_silu_mean(const void* x, void* y, const float* w9c, const float* lnw, const float* lnb, int32_t F, int32_t H, int32_t W,
                                   int32_t C, float eps, float* mean, void* ws, int64_t ws_bytes, void* stream);
/* SE excite + scale in one launch: x[f, :, c] *= sigmoid(W2[c, :rd] . g1[f, :rd] + b2[c])  (= vl2_small_linear(.., SIGMOID) then vl2_se_scale).
 * g1 fp32 [F, rd] (the squeezed, reduced, SiLU'd vector), W2 [C, rd] in the element type, b2 fp32 [C] or NULL.  C%8==0, rd%16==0. */
int32_t vl2_se_excite_scale(void* x, const float* g1, const void* W2, const float* b2, int32_t F, int32_t HW, int32_t C, int32_t rd, void* stream);

/* Mistral RoPE (rotate-half) on q,k of the fused qkv rows + KV-cache append at positions pos0..pos0+S-1.
 * head_dim 128.  cos/sin fp32 [maxpos][64].  HF:modeling_mistral.py apply_rotary_pos_emb, DynamicCache.update. */
int32_t vl2_rope_kv(const void* qkv, void* q_out, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                    int32_t S, int32_t nh, int32_t nkv, int32_t smax, int32_t pos0, void* stream);

/* y[N] = W[N,K] x[K] (+ bias[N]) (+ res[N]) for one token (decode).  norm_w != NULL fuses MistralRMSNorm on x first;
 * bias (fp32, may be NULL) is Qwen2's q/k/v bias (HF:models/qwen2/modeling_qwen2.py Qwen2Attention).  flags as vl2_gemm_desc.flags. */
int32_t vl2_gemv_bf16(const void* W, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                      int32_t N, int32_t K, int32_t ldw, float eps, int32_t flags, void* stream);
/* ---- fp8 weights for decode (the fp8 half of BASELINE.json configs[4]; csrc/k_fp8.h).  OCP e4m3fn bytes (gfx950's format,
 * not MI300X's fnuz) with ONE power-of-two fp32 scale per output row: W[n][k] ~= scale[n] * q[n][k]; activations stay 16-bit (W8A16), fp32
 * products and sums.  The reference has no fp8 path: the quantiser IS the definition, restated in oracle/fp8_oracle.py.
 * vl2_pack_quant_fp8: w [N, ldw] 16-bit elements (K % 16 == 0) -> q [N, K] bytes, scale [N]: scale = 2^e, e the smallest integer with
 *   max|row| <= 448 * 2^e (1 for a zero row), q = round-to-nearest-even e4m3fn of w * 2^-e (exact scaling: reproducible bit for bit anywhere).
 * vl2_gemv_fp8: y[N] = scale[N] * (q[N,K] x[K]) (+ bias) (+ res), one token; norm_w / VL2_GEMV_RMS_PLAIN / VL2_GEMM_SWIGLU /
 *   VL2_GEMM_OUT_F32 as vl2_gemv_bf16 (SWIGLU: q and scale in the packed 64-row block order).  N even, K % 16 == 0, K <= 32704. */
int32_t vl2_pack_quant_fp8(const void* w, int64_t N, int64_t K, int64_t ldw, void* q, float* scale, void* stream);
/* fp8 ACTIVATIONS for vl2_gemm's VL2_GEMM_FP8 form (the prefill on the fp8 matrix pipe): x [M, ldx] 16-bit elements -> q [M, ldq] e4m3fn bytes with the
 * power-of-two row scale rule of vl2_pack_quant_fp8, and row_tab [M][2] = (0, sa[m]) (norm = VL2_NORM_NONE) or (0, sa[m] * rsqrt(mean_k x^2 + eps))
 * (VL2_NORM_RMS: the GEMM then computes RMSNorm(x) W'^T with the norm weight folded into W', HF:modeling_mistral.py:46-48) -- pass it as
 * vl2_gemm_desc.row_norm.  K % 16 == 0.  Definition restated in oracle/fp8_oracle.py (quant_rows / gemm_w8a8). */
int32_t vl2_quant_act_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* row_tab, int32_t M, int32_t K, int32_t norm, float eps, void* stream);
int32_t vl2_gemv_fp8(const void* q, const float* scale, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                     int32_t N, int32_t K, int32_t ldq, float eps, int32_t flags, void* stream);
/* Skinny-M GEMM for batched decode: C[M <= 64, N] = A[M,K] W[N,K]^T (+bias | +res | SwiGLU | fp32 out; flags as vl2_gemm_desc.flags).
 * The weights stream from HBM straight into the B operand of v_mfma_f32_16x16x32_bf16 (GEMV-style, once for all M rows), K is
 * split over workgroups, fp32 partial sums go through the caller's workspace `ws` (required, >= vl2_workspace_bytes()) and are reduced in order. */
int32_t vl2_gemm_skinny_bf16(const void* A, const void* W, void* C, const float* bias, const void* res, int32_t M, int32_t N,
                             int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t ldres, int32_t flags, void* ws, int64_t ws_bytes,
                             void* stream);
/* Batched decode: y[b][N] = W[N,K] x[b][K] (+ bias) (+ res[b]) for MB sequences in ONE pass over W (up to
 * 4 rows per launch, as many as fit 64 KiB of LDS; larger MB is split).  ldx / ldy / ldres = element strides between rows.
 * Same fused RMSNorm / bias / residual / SwiGLU semantics as vl2_gemv_bf16, applied per row. */
int32_t vl2_gemv_batched_bf16(const void* W, const void* x, const float* norm_w, const void* res, const float* bias, void* y,
                              int32_t MB, int32_t N, int32_t K, int32_t ldw, int32_t ldx, int32_t ldy, int32_t ldres, float eps,
                              int32_t flags, void* stream);
/* Decode attention for ONE new token at position pos, fused with RoPE and the KV-cache append:
 *   qkv [(nh+2*nkv)*128] = un-roped fused q|k|v projection of the token; the kernel ropes q, ropes k and appends k,v to
 *   cache row pos (HF apply_rotary_pos_emb + DynamicCache.update), then softmax(q K^T / sqrt(d)) V over rows [0, pos]
 *   split in 64-key slices (flash-decoding), and combines the slices into out [nh*128] bf16.
 *   pos_dev != NULL: the position is read from device memory (*pos_dev) so a captured hipGraph replays as it moves; the
 *   launch then covers positions < ctx_cap.  partial: fp32 workspace >= nh*ceil(cap/64)*130 floats (cap = ctx_cap or pos+1). */
#ifdef VL2_EXPERIMENTAL
/* ---- LAB ENTRY POINTS (not permanent ABI): exported by libvl2hip_lab.so only (-DVL2_LAB, scripts/build_lab_lib.sh), for the A/B tests and the lab
 * scripts that measured them SLOWER than the launches they replace.  (Their stage flags: VL2_STAGE_FUSED_DECODE_ATTN, VL2_STAGE_DECODE_TAIL.) */
/* The same attention + combine in ONE launch (the workgroup that finishes a kv head's last slice combines its q heads): position from
 * device memory only, partial sized for smax (nh*ceil(smax/64)*130 floats), cnt = nkv int32 ticket counters that must be ZERO when the
 * launch starts (the caller clears them; NOT with a hipMemsetAsync node of a few bytes inside a captured hipGraph -- that did not replay
 * on ROCm 7.2).  Same bits as vl2_attn_decode; measured slower than it on MI355X (profiles/r03_experiments.md section 5), so nothing in
 * the product takes it by default. */
int32_t vl2_attn_decode_fused(const void* qkv, void* kcache, void* vcache, const float* cos_t, const float* sin_t, float* partial,
                              void* out, int32_t nh, int32_t nkv, int32_t smax, const int32_t* pos_dev, float scale, int32_t* cnt,
                              void* stream);
/* Decode "tail engine" (csrc/k_decode_tail.h): x1 = x0 + Wo o; act = silu(gate) * up of RMSNorm(x1) (norm weight folded into Wgu, packed as
 * for VL2_GEMM_SWIGLU); xout = x1 + Wd act -- HF:modeling_mistral.py:229-241 for one token -- as ONE persistent launch (one 1024-thread
 * workgroup per CU, two grid barriers, the next phase's first weight rows in flight across each barrier) instead of three vl2_gemv_bf16
 * launches; the same bits as those.  o [QD], x0 / x1 / xout [D], act [I] bf16 (xout may be x0); Wo [D, QD], Wgu [2 I, D], Wd [D, I].
 * bar: 32 int32 words that must be ZERO when the launch starts (it re-arms them; vl2_llm_decode_step clears them in its argmax launch).
 * Every spin is bounded: on a timeout bar[24] is set and the outputs are garbage (the caller must read bar[24]; vl2_llm_decode_step does not).
 * D, QD, I <= 32512 (the staged vector shares the 64 KiB of LDS a kernel may take without the opt-in attribute with 512 B of static LDS), multiples of 8 (I of 32).
 * MEASURED SLOWER than the three launches on MI355X (scripts/ubench/tail_lab.hip: 98.5 vs 66.7 us per Mistral-7B layer -- a grid barrier
 * with its write-through hand-off costs ~11 us against ~3.6 us of fixed cost per GEMV launch, and one 16-wave workgroup per CU streams the
 * gate/up rows at 5 TB/s instead of 6.4): nothing takes it by default (VL2_STAGE_DECODE_TAIL). */
int32_t vl2_decode_tail(const void* Wo, const void* Wgu, const void* Wd, int32_t ldwo, int32_t ldwgu, int32_t ldwd, const void* o, const void* x0,
                        void* x1, void* act, void* xout, int32_t D, int32_t QD, int32_t I, float eps, int32_t* bar, void* stream);
#endif /* VL2_EXPERIMENTAL */
int32_t vl2_attn_decode(const void* qkv, void* kcache, void* vcache, const float* cos_t, const float* sin_t, float* partial,
                        void* out, int32_t nh, int32_t nkv, int32_t smax, int32_t pos, const int32_t* pos_dev,
                        int32_t ctx_cap, float scale, void* stream);
/* Batched form: B sequences, each with its own KV cache (kcache + b*cache_bs ...), fused qkv row (qkv + b*qkv_bs), output row
 * (out + b*out_bs) and position pos_dev[b] (device int32[B]; the launch covers positions < ctx_cap).  partial must hold
 * B * nh * ceil(ctx_cap/64) * 130 floats. */
int32_t vl2_attn_decode_batched(const void* qkv, void* kcache, void* vcache, const float* cos_t, const float* sin_t, float* partial,
                                void* out, int32_t B, int64_t qkv_bs, int64_t cache_bs, int64_t out_bs, int32_t nh, int32_t nkv,
                                int32_t smax, const int32_t* pos_dev, int32_t ctx_cap, float scale, void* stream);
/* greedy argmax (first maximal index) of fp32 logits -> *tok (device int32) and hist[step] if hist != NULL.
 * state != NULL (device int32[2] = {position, step}): hist index = state[1], then both counters advance by one, so the
 * whole decode step is replayable from a hipGraph.  HF:generation/utils.py _sample with do_sample=False. */
int32_t vl2_argmax(const float* logits, int32_t V, int32_t* tok, int32_t* hist, int32_t step, int32_t* state, void* stream);
/* do_sample=True (videollama2/__init__.py:93-106 -> HF GenerationMixin._sample): the logits warpers of HF:generation/logits_process.py in HF's order --
 * scores / temperature; top_k > 0: scores below the k-th largest removed (ties stay; HF's generation_config default is 50); top_p < 1: ascending
 * softmax cumsum <= 1 - top_p removed, the largest always stays -- then ONE draw from softmax(kept scores): the token whose interval of the cumulative
 * distribution, taken in token-index order, contains u[step] (u in [0, 1), device fp32; with `state` the index is state[1]).  torch.multinomial's own
 * consumption of its random stream is not reproducible outside torch, so parity = the kept set and the probabilities (oracle/sampling_oracle.py,
 * pinned to the live HF warpers).  tok / hist / step / state exactly as vl2_argmax (the call takes its place in a decode loop or captured graph).
 * dbg: optional 4 floats {tokens kept, kept mass / top-k mass, threshold score, largest score}.  One 1024-thread workgroup, no sort (radix descent over
 * the float keys; probability mass in 2^-40 fixed point: deterministic). */
int32_t vl2_sample_token(const float* logits, int32_t V, float temperature, int32_t top_k, float top_p, const float* u, int32_t* tok, int32_t* hist,
                         int32_t step, int32_t* state, float* dbg, void* stream);
/* out[i,:] = table[ids[i],:]; ids int32 device.  embed_tokens in videollama2/model/videollama2_arch.py:203-220. */
int32_t vl2_embed_rows(const int32_t* ids, const void* table, void* out, int32_t n, int32_t D, int32_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------------------------------------
 * Stage-level entry points: a whole stage of the hot path in ONE call.  The host loop over the layers runs
 * inside the library (C++), on the same kernels and through the same argument checks as the per-operator entry points above; a
 * non-Python host (or a hipGraph capture) drives the path with a handful of calls instead of several hundred.  Same rules: no
 * allocation, no synchronisation, everything enqueued on `stream`; activations live in a caller-owned workspace whose size the
 * matching *_workspace_bytes query returns.  The weight structs hold DEVICE pointers to tensors in the packed layouts of
 * videollama2_amd/weights.py (nn.Linear [N, K] bf16, fused q|k|v, gate/up interleaved for VL2_GEMM_SWIGLU, norm weights folded
 * into the following projection -- weights.fold_norm); the `layers` arrays themselves are HOST arrays. */

/* CLIP / SigLIP vision tower: frames -> hidden_states[select_layer] (videollama2/model/encoder.py:41-53, :111-123 -> HF
 * CLIPVisionModel / SiglipVisionModel forward up to the selected layer). */
typedef struct vl2_vit_layer {
    const void* wqkv; const float* bqkv; const float* sqkv;      /* layer_norm1 folded: W', W b + c, column sums of W' */
    const void* wo;   const float* bo;
    const void* w1;   const float* b1;   const float* s1;        /* layer_norm2 folded into fc1 */
    const void* w2;   const float* b2;
} vl2_vit_layer;
typedef struct vl2_vit_desc {
    uint32_t size;                 /* sizeof(vl2_vit_desc) */
    int32_t family;                /* 0 = CLIP (class token + pre_layrnorm, no patch bias), 1 = SigLIP (patch bias, neither) */
    int32_t image, patch, D, I, heads, head_dim, n_layers, kp;   /* I / head_dim as PACKED (SigLIP: zero-padded), kp = padded 3*patch^2 */
    int32_t act;                   /* VL2_ACT_QGELU (CLIP) / VL2_ACT_GELU_TANH (SigLIP) */
    float eps, attn_scale;
    const void* patch_w; const float* patch_b; const void* pos; const void* cls_pos; const float* pre_w; const float* pre_b;
    const vl2_vit_layer* layers;   /* host array [n_layers] */
    uint32_t flags;                /* VL2_STAGE_* */
} vl2_vit_desc;
int64_t vl2_vit_workspace_bytes(const vl2_vit_desc* w, int32_t T);
/* frames: frame_dtype 0 fp32 / 1 fp16 / 2 bf16 [T,3,image,image], or 3 = uint8 [T,image,image,3] normalised in the patch-row
 * kernel with (rescale, mean_rgb[3], std_rgb[3]) (host array of 7 floats, read during the call).  out: bf16
 * [T * (patches + (family == 0)), D], the class-token row included (the caller drops it for select_feature 'patch'). */
int32_t vl2_vit_forward(const vl2_vit_desc* w, const void* frames, int32_t frame_dtype, const float* u8_norm7, int32_t T, void* out,
                        void* ws, int64_t ws_bytes, void* stream);

/* ---- one-time weight re-layout (checkpoint tensors, bf16, device memory -> the layouts the entry points above read).  What
 * videollama2_amd/weights.py does with tensor ops, for hosts without PyTorch; csrc/k_pack.h.  All pointers 16-byte aligned.
 * vl2_pack_fold_norm: LayerNorm / RMSNorm affine of HF CLIPEncoderLayer.layer_norm1/2, MistralDecoderLayer.input_layernorm /
 *   post_attention_layernorm folded into the linear layer that follows: Wp = bf16(W * g) per input column, colsum[n] = sum_k Wp[n][k],
 *   shift[n] = sum_k W[n][k] * beta[k] + bias[n] (beta / bias / shift may be NULL: RMSNorm).  Feeds vl2_gemm_desc.w_colsum / bias.
 * vl2_pack_gate_up: MistralMLP gate_proj / up_proj [I, D] -> [2I, D], blocks of 64 rows = 32 gate rows then 32 up rows (VL2_GEMM_SWIGLU).
 * vl2_pack_permute: [A][B][C] -> [A][C][B]; STC sampler Conv3d weight [Co][Ci][2*2*2] -> [Co][tap][Ci] (A=Co, B=Ci, C=8), depthwise
 *   3x3 [C][9] -> [9][C] with out_f32 = 1 (A=1, B=C, C=9).
 * vl2_pack_pad_rows: rows x cols_src -> rows x cols_dst, zero filled (patch weight K 588 -> 640; SigLIP head_dim 72 -> 96 and MLP
 *   4304 -> 4352: a "row" is the block being padded).   vl2_pack_cvt_f32: bf16 -> fp32 (norm weights, biases). */
int32_t vl2_pack_fold_norm(const void* W, const void* g, const void* beta, const void* bias, void* Wp, float* colsum, float* shift,
                           int32_t N, int32_t K, int32_t ldw, void* stream);
int32_t vl2_pack_gate_up(const void* gate, const void* up, void* out, int32_t I, int32_t D, void* stream);
int32_t vl2_pack_permute(const void* in, void* out, int32_t A, int32_t B, int32_t C, int32_t out_f32, void* stream);
int32_t vl2_pack_pad_rows(const void* in, void* out, int64_t rows, int64_t cols_src, int64_t cols_dst, void* stream);
int32_t vl2_pack_cvt_f32(const void* in, float* out, int64_t n, void* stream);

/* STC connector (videollama2/model/projector.py:189-215; timm RegStage Bottleneck x 4, Conv3d sampler, RegStage x 4, readout MLP):
 * tower features x [T * hw * hw, cin] bf16 (token-major = channels-last) -> visual tokens out [To * Ho * Wo, C] bf16. */
typedef struct vl2_stc_block {
    const void* conv1_w; const float* bn1_w; const float* bn1_b;
    const float* dw_w;   const float* bn2_w; const float* bn2_b;     /* depthwise 3x3 taps [9][C] fp32 */
    const void* fc1_w; const float* fc1_b; const void* fc2_w; const float* fc2_b;   /* SE: [rd, C], [C, rd] */
    const void* conv3_w; const float* bn3_w; const float* bn3_b;
    const void* ds_w; const float* dsbn_w; const float* dsbn_b;      /* 1x1 shortcut (first block of s1 only) or NULL */
    int32_t rd;                                                      /* SE reduction width */
} vl2_stc_block;
typedef struct vl2_stc_desc {
    uint32_t size;
    int32_t cin, C;                /* tower width, connector / LLM width */
    vl2_stc_block s1[4], s2[4];
    const void* samp_w; const float* samp_b;                         /* Conv3d as [C, 8*C], K order (kt, kh, kw, cin) */
    const void* ro0_w; const float* ro0_b; const void* ro2_w; const float* ro2_b;
    uint32_t flags;                /* VL2_STAGE_* */
} vl2_stc_desc;
int64_t vl2_stc_workspace_bytes(const vl2_stc_desc* w, int32_t T, int32_t hw, int32_t n_out);
/* conv3d_idx: device int32 [8][n_out] gather table of the Conv3d(k2, s2, padding 1 | 0) taps (row of the s1 output per tap and
 * output position, -1 = zero padding; videollama2_amd/connector.py conv3d_k2s2p1_index), n_out = To*Ho*Wo. */
int32_t vl2_stc_forward(const vl2_stc_desc* w, const void* x, int32_t T, int32_t hw, const int32_t* conv3d_idx, int32_t To, int32_t Ho,
                        int32_t Wo, void* out, void* ws, int64_t ws_bytes, void* stream);

/* Mistral / Qwen2 decoder (HF:models/mistral/modeling_mistral.py MistralModel.forward + lm_head; Qwen2: q/k/v bias). */
typedef struct vl2_llm_layer {
    const void* wqkv; const float* bqkv;     /* input_layernorm folded into the columns; bqkv NULL for Mistral */
    const void* wo;
    const void* wgu;                         /* post_attention_layernorm folded; gate/up interleaved (VL2_GEMM_SWIGLU) */
    const void* wd;
    void* kcache; void* vcache;              /* [kv_heads][smax][128] bf16 */
} vl2_llm_layer;
typedef struct vl2_llm_layer_w8 {            /* fp8 copies of a layer's packed projections (vl2_pack_quant_fp8 of wqkv / wo / wgu / wd) */
    const void* wqkv; const float* sqkv;
    const void* wo;   const float* so;
    const void* wgu;  const float* sgu;
    const void* wd;   const float* sd;
} vl2_llm_layer_w8;
typedef struct vl2_llm_desc {
    uint32_t size;
    int32_t D, I, heads, kv_heads, n_layers, vocab, smax;        /* head_dim is 128; I as packed */
    float eps;
    const vl2_llm_layer* layers;   /* host array [n_layers] */
    const void* embed; const float* norm_w; const float* ones /* [D] of 1.0f */; const void* lm_head;
    const float* cos_t; const float* sin_t;                      /* fp32 [smax][64] */
    uint32_t flags;                /* VL2_STAGE_* */
    const vl2_llm_layer_w8* layers_w8;                            /* host array [n_layers] or NULL; read only with VL2_STAGE_DECODE_FP8 / VL2_STAGE_PREFILL_FP8 */
    const void* lm_head_w8; const float* lm_head_scale;          /* fp8 copy of lm_head (final norm weight NOT folded: norm_w is applied) */
} vl2_llm_desc;
int64_t vl2_llm_workspace_bytes(const vl2_llm_desc* w, int32_t S);   /* for `w->flags` as set: the fp8 activation image is part of it only with VL2_STAGE_PREFILL_FP8 */
/* Prefill: inputs_embeds x [S, D] bf16 -> K/V cache rows 0..S-1 of every layer, fp32 logits of the LAST position [vocab]. */
int32_t vl2_llm_prefill(const vl2_llm_desc* w, const void* x, int32_t S, float* logits_last, void* ws, int64_t ws_bytes, void* stream);
/* One greedy decode step, hipGraph-replayable (everything that moves lives on the device): tok = argmax(logits) (also written
 * to hist[state[1]]; state = {position, step} advance by one), then the token's forward at position state[0] -> logits of the
 * next step (in place).  partial: fp32 [heads * ceil(smax/64) * 130].  HF:generation/utils.py _sample, do_sample=False. */
int32_t vl2_llm_decode_step(const vl2_llm_desc* w, float* logits, int32_t* tok, int32_t* state, int32_t* hist, float* partial,
                            void* ws, int64_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif

// Row normalisations (HBM-bound; one wave per row, 16-B lane loads, fp32 statistics, wave-shuffle reductions).
//   layernorm_kernel: nn.LayerNorm over the last dim, eps inside the sqrt           (HF:models/clip/modeling_clip.py
//                     CLIPEncoderLayer.layer_norm1/2, pre_layrnorm; timm LayerNormAct2d in channels-last form for the
//                     STC connector, videollama2/model/projector.py:153-184) with optional "+ residual" and SiLU
//                     (the tail of timm Bottleneck: act3(conv3(x) + shortcut)).
//   rmsnorm_kernel:   HF:models/mistral/modeling_mistral.py MistralRMSNorm.forward (fp32 upcast, rsqrt(mean x^2 + eps)).
#pragma once
#include "dev_common.h"

struct NormArgs {
    const bf16_t* x;      // [rows, ldx]
    bf16_t* y;            // [rows, ldy]
    const float* w;       // [C]
    const float* b;       // [C] (layernorm) or null
    const bf16_t* res;    // [rows, ldres] added AFTER the affine (or null)
    int rows, C, ldx, ldy, ldres;
    float eps;
    int silu;             // apply SiLU last
};

// NV = number of 8-element vectors per lane (C <= NV*512)
template <int NV, bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(NormArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const bf16_t* x = p.x + (size_t)row * p.ldx;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < p.C) {
            unpack8(*(const u32x4*)(x + c), v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += RMS ? v[i][j] * v[i][j] : v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    s = wave_sum(s);
    const float inv_c = 1.0f / (float)p.C;
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s * inv_c + p.eps);
    } else {
        mean = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < p.C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q * inv_c + p.eps);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < p.C) {
            const f32x4 w0 = *(const f32x4*)(p.w + c), w1 = *(const f32x4*)(p.w + c + 4);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * (j < 4 ? w0[j] : w1[j - 4]);
            if (!RMS && p.b) {
                const f32x4 b0 = *(const f32x4*)(p.b + c), b1 = *(const f32x4*)(p.b + c + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += (j < 4 ? b0[j] : b1[j - 4]);
            }
            if (p.res) {
                float r[8];
                unpack8(*(const u32x4*)(p.res + (size_t)row * p.ldres + c), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
            if (p.silu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = silu_f(o[j]);
            }
            *(u32x4*)(p.y + (size_t)row * p.ldy + c) = pack8(o);
        }
    }
}

// Wide form for FEW, LONG rows (the LLM prefill: 1621 rows x 4096): one workgroup per row, 256 lanes x NVW (2: C <= 4096,
// 4: C <= 8192, the 72B decoder) vectors,
// so the row count, not rows/4, is the number of workgroups in flight (one wave per row leaves a 256-CU chip with ~1.6
// workgroups per CU: 13 us for 26 MB).  Same per-element arithmetic; the row sums are reduced wave-then-LDS instead of in
// one wave, i.e. in a different fp32 order than norm_kernel -- the launcher picks it by C alone (never by the row count) and
// only for RMSNorm, so a row's result depends neither on how many rows are normalised with it (batched prefill == one by
// one) nor, for the per-frame LayerNorms, on how many frames a rank holds.
template <bool RMS, int NVW>
__global__ __launch_bounds__(256) void norm_wide_kernel(NormArgs p) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    const bf16_t* x = p.x + (size_t)row * p.ldx;
    float v[NVW][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < p.C) {
            unpack8(*(const u32x4*)(x + c), v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += RMS ? v[i][j] * v[i][j] : v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    auto block_sum = [&](float t) {
        t = wave_sum(t);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    s = block_sum(s);
    const float inv_c = 1.0f / (float)p.C;
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s * inv_c + p.eps);
    } else {
        mean = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NVW; ++i) {
            const int c = (i * 256 + threadIdx.x) * 8;
            if (c < p.C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        }
        q = block_sum(q);
        rstd = rsqrtf(q * inv_c + p.eps);
    }
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < p.C) {
            const f32x4 w0 = *(const f32x4*)(p.w + c), w1 = *(const f32x4*)(p.w + c + 4);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * (j < 4 ? w0[j] : w1[j - 4]);
            if (!RMS && p.b) {
                const f32x4 b0 = *(const f32x4*)(p.b + c), b1 = *(const f32x4*)(p.b + c + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += (j < 4 ? b0[j] : b1[j - 4]);
            }
            if (p.res) {
                float r[8];
                unpack8(*(const u32x4*)(p.res + (size_t)row * p.ldres + c), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
            if (p.silu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = silu_f(o[j]);
            }
            *(u32x4*)(p.y + (size_t)row * p.ldy + c) = pack8(o);
        }
    }
}

// (sum, sum of squares) of every 64-column block of every row of x [rows, C] -> stats [rows][C/64][2] fp32: the statistics a
// norm-carrying GEMM consumes (k_gemm.h `gemm_row_stats`), in the layout AND the summation order of the GEMM epilogue that
// normally emits them (8 consecutive values per lane in sequence, then `octet_sum` over the 8 lanes of a block),
// so a row's statistics are the same bits whether this kernel or a producer GEMM wrote them.  One wave per row; a pass covers
// 512 columns (8 blocks).  Seeds the chain for tensors no GEMM wrote: inputs_embeds, the CLIP embeddings after pre_layrnorm.
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ stats, int rows, int C, int ldx) {
#pragma clang fp reassociate(off)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                               // wave-uniform
    const bf16_t* xr = x + (size_t)row * ldx;
    const int np = C >> 6;
    for (int c0 = 0; c0 < C; c0 += 512) {
        const int c = c0 + lane * 8;
        float s = 0.f, q = 0.f;
        if (c < C) {
            float v[8];
            unpack8(*(const u32x4*)(xr + c), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s += v[j]; q = __builtin_fmaf(v[j], v[j], q); }
        }
        s = octet_sum(s);
        q = octet_sum(q);
        if (c < C && (lane & 7) == 0) {
            float* dst = stats + ((size_t)row * np + (c >> 6)) * 2;
            dst[0] = s;
            dst[1] = q;
        }
    }
}

// The K/64 partial (sum, sum of squares) of every row -> (mean, rstd): the reduction `gemm_row_stats` (k_gemm.h) would
// otherwise repeat in every column tile of the consuming GEMM (112 times for the gate/up projection).  Eight lanes per row:
// lane i of the octet sums partials i, i + 8, ... (coalesced 8-byte loads), `octet_sum` folds the eight -- a fixed order.
// kind: 1 RMSNorm (mean = 0), 2 LayerNorm.  (First version: one thread per row, 5.8 us per call on MI355X -- a latency chain
// of up to 32 dependent loads on 7 workgroups; this form keeps ~50-290 workgroups busy with 2-8 loads per lane.)
__global__ __launch_bounds__(256) void row_norm_finalize_kernel(const float* __restrict__ stats, float* __restrict__ out, int rows, int np,
                                                                int K, int kind, float eps) {
#pragma clang fp reassociate(off)
    const int m = blockIdx.x * 32 + (threadIdx.x >> 3), i8 = threadIdx.x & 7;
    const int mc = m < rows ? m : rows - 1;                    // whole octets stay active for the DPP reduction
    const float* sp = stats + (size_t)mc * np * 2;
    float sum = 0.f, sq = 0.f;
    for (int i = i8; i < np; i += 8) {
        const f32x2 v = *(const f32x2*)(sp + 2 * i);
        sum += v[0];
        sq += v[1];
    }
    sum = octet_sum(sum);
    sq = octet_sum(sq);
    if (m < rows && i8 == 0) {
        const float inv = 1.0f / (float)K;
        float mean = 0.f, rstd;
        if (kind == 2) {
            mean = sum * inv;
            rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, sq * inv), 0.f) + eps);
        } else {
            rstd = rsqrtf(sq * inv + eps);
        }
        out[2 * (size_t)m] = mean;
        out[2 * (size_t)m + 1] = rstd;
    }
}

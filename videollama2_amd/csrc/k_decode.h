// Mistral decoder glue + single-token (decode) kernels.  HBM-bound: the per-token cost is streaming the bf16 weights.
//   rope_kv_kernel           : rotate-half RoPE on q,k (HF:models/mistral/modeling_mistral.py apply_rotary_pos_emb /
//                              rotate_half) from the fused qkv GEMM output, + KV-cache append (DynamicCache.update).
//   gemv_bf16_kernel         : y = W x for one token (q/k/v/o/gate/up/down/lm_head at M=1), 16-B lane loads straight to
//                              VGPRs, v_dot2c_f32_bf16, optional fused RMSNorm prologue, residual / SwiGLU epilogue.
//   attn_decode_kernel (+combine): RoPE + KV append + one query token against the KV cache, split over the context.
//   argmax_kernel            : greedy token (HF:generation/utils.py _sample, do_sample=False -> torch.argmax).
//   embed_rows_kernel        : embed_tokens gather (videollama2/model/videollama2_arch.py:203-220).
#pragma once
#include "dev_common.h"

// c + a.lo * b.lo + a.hi * b.hi of two packed element pairs (v_dot2c_f32_bf16 / v_dot2_f32_f16)
#ifdef VL2_ELEM_F16
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(vl2_f16x2, a), __builtin_bit_cast(vl2_f16x2, b), c, false);
}
#else
typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2_t, a), __builtin_bit_cast(bf16v2_t, b), c, false);
}
#endif

// rotate-half RoPE of one (x[d], x[d + 64]) pair, the contraction pinned (shared by the decode kernels, which must agree to the bit)
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float sn, float& o1, float& o2) {
    o1 = __builtin_fmaf(x1, c, -(x2 * sn));       // q*cos + rotate_half(q)*sin, first half:  -x2
    o2 = __builtin_fmaf(x2, c, x1 * sn);          //                              second half: +x1
}
// qkv [S, (nh+2*nkv)*HD] -> q_out [S, nh*HD] (roped), kcache/vcache [nkv][smax][HD] rows pos0+s.
// cos/sin: fp32 [maxpos][HD/2].  One thread per (token, head, 8-dim chunk of the first half).  HD = 128.
__global__ __launch_bounds__(256) void rope_kv_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ q_out,
                                                      bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                      int S, int nh, int nkv, int smax, int pos0) {
    constexpr int HD = 128, HALF = 64;
    const int heads = nh + 2 * nkv;
    const size_t total = (size_t)S * heads * 8;
    for (size_t it = (size_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
        const int ch = (int)(it & 7);
        const int head = (int)((it >> 3) % heads);
        const int s = (int)(it / ((size_t)heads * 8));
        const bf16_t* src = qkv + (size_t)s * heads * HD + head * HD + ch * 8;
        const u32x4 a = *(const u32x4*)src, b = *(const u32x4*)(src + HALF);
        const int pos = pos0 + s;
        if (head < nh + nkv) {
            float x1[8], x2[8], o1[8], o2[8];
            unpack8(a, x1); unpack8(b, x2);
            const float* cp = cos_t + (size_t)pos * HALF + ch * 8;
            const float* sp = sin_t + (size_t)pos * HALF + ch * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float c = cp[j], sn = sp[j];
                o1[j] = x1[j] * c - x2[j] * sn;       // q*cos + rotate_half(q)*sin, first half:  -x2
                o2[j] = x2[j] * c + x1[j] * sn;       //                              second half: +x1
            }
            bf16_t* dst = head < nh ? q_out + (size_t)s * nh * HD + head * HD + ch * 8
                                    : kcache + ((size_t)(head - nh) * smax + pos) * HD + ch * 8;
            *(u32x4*)dst = pack8(o1);
            *(u32x4*)(dst + HALF) = pack8(o2);
        } else {
            bf16_t* dst = vcache + ((size_t)(head - nh - nkv) * smax + pos) * HD + ch * 8;
            *(u32x4*)dst = a;
            *(u32x4*)(dst + HALF) = b;
        }
    }
}

struct GemvArgs {
    const bf16_t* W;        // [N, ldw]  (SWIGLU: packed blocks of 64 rows = 32 gate rows then 32 up rows)
    const bf16_t* x;        // [K]
    const float* norm_w;    // fused RMSNorm prologue on x (or null)
    const bf16_t* res;      // [N_out] residual (or null)
    void* y;                // bf16 or fp32 [N_out]
    int N, K, ldw;
    float eps;
    const float* bias;      // [N_out] added before the residual (Qwen2 q/k/v bias) or null; not with SWIGLU
    int ldx, ldy, ldres;    // gemv_mr only: element strides between the MB rows of x / y / res
    int rms_plain;          // gemv_bf16_kernel: RMS-normalise x without a weight vector (norm_w folded into W): (v * rstd) * 1 == v * rstd, same bits
};

// grid = ceil(N_out / (4*RPW)), block 256; dynamic LDS = K * 2 bytes (x as bf16)
template <bool SWIGLU, bool OUT_F32, int RPW>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(GemvArgs p) {
    // The fused RMSNorm must give the SAME bits here and in gemv_mr_bf16_kernel (a row of a batched decode step == the
    // single-sequence step): under -ffast-math the two instantiations were free to associate "v * rstd * w" and the
    // sum of squares differently (seen on hardware: one logit row in thousands off by 3e-4).  Fixed order, explicit FMAs.
#pragma clang fp reassociate(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    __shared__ float red[8];
    bf16_t* xs = (bf16_t*)vl2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_out = SWIGLU ? p.N / 2 : p.N;
    const int nvec = p.K >> 3;                       // 16-B vectors per row
    // K <= 4096 (every projection except down_proj): a row is ONE pass of 8 loads per lane.  The weights do not depend on
    // x, so the first row's loads are issued BEFORE x is normalised and staged: the two memory latencies overlap instead
    // of adding up (these GEMVs are a single round trip long: 11.9 -> ~9 us for the 50 MB qkv projection).
    const bool one_pass = nvec <= 512;
    u32x4 wv[8], uv[8];
    auto issue_row = [&](int j, int v0) {
        const int row0 = SWIGLU ? (j >> 5) * 64 + (j & 31) : j;
        const bf16_t* w0p = p.W + (size_t)row0 * p.ldw;
        const bf16_t* w1p = w0p + (size_t)32 * p.ldw;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = v0 + i * 64 + lane;
            if (v < nvec) {
                wv[i] = __builtin_nontemporal_load((const u32x4*)(w0p + (size_t)v * 8));
                if (SWIGLU) uv[i] = __builtin_nontemporal_load((const u32x4*)(w1p + (size_t)v * 8));
            }
        }
    };
    const int jfirst = (blockIdx.x * 4 + wave) * RPW;
    if (one_pass && jfirst < n_out) issue_row(jfirst, 0);
    // stage x (optionally RMS-normalised: HF MistralRMSNorm, fp32 statistics, result rounded to bf16)
    float rstd = 1.f;
    const bool norm = p.norm_w != nullptr || p.rms_plain;
    if (norm) {
        float ss = 0.f;
        for (int k = tid * 8; k < p.K; k += 2048) {
            float v[8];
            unpack8(*(const u32x4*)(p.x + k), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(v[j], v[j], ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)p.K + p.eps);
    }
    for (int k = tid * 8; k < p.K; k += 2048) {
        u32x4 raw = *(const u32x4*)(p.x + k);
        if (norm) {
            float v[8];
            unpack8(raw, v);
            f32x4 w0 = {1.f, 1.f, 1.f, 1.f}, w1 = w0;
            if (p.norm_w) { w0 = *(const f32x4*)(p.norm_w + k); w1 = *(const f32x4*)(p.norm_w + k + 4); }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (v[j] * rstd) * (j < 4 ? w0[j] : w1[j - 4]);
            raw = pack8(v);
        }
        *(u32x4*)(xs + k) = raw;
    }
    __syncthreads();

#pragma unroll 1
    for (int r = 0; r < RPW; ++r) {
        const int j = jfirst + r;
        if (j >= n_out) break;
        float a0 = 0.f, a1 = 0.f;
        for (int v0 = 0; v0 < nvec; v0 += 64 * 8) {
            if (!(one_pass && r == 0)) issue_row(j, v0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int v = v0 + i * 64 + lane;
                if (v < nvec) {
                    const u32x4 xv = *(const u32x4*)(xs + (size_t)v * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a0 = dot2_bf16(wv[i][q], xv[q], a0);
                        if (SWIGLU) a1 = dot2_bf16(uv[i][q], xv[q], a1);
                    }
                }
            }
        }
        a0 = wave_sum(a0);
        if (SWIGLU) a1 = wave_sum(a1);
        if (lane == 0) {
            float o = SWIGLU ? silu_f(a0) * a1 : a0;
            if (!SWIGLU && p.bias) o += p.bias[j];
            if (p.res) o += bf2f(p.res[j]);
            if (OUT_F32) ((float*)p.y)[j] = o;
            else ((bf16_t*)p.y)[j] = f2bf(o);
        }
    }
}

// gemv_bf16_kernel with the activation vector requested BEFORE the weight row: a wave's loads return in order, so x queued behind the row's
// eight 16-B weight loads per lane cannot be normalised and staged before they have landed -- requested first, the RMSNorm prologue (two
// barriers, an LDS round trip) runs while the weights are still in flight.  Same arithmetic in the same order: the same bits.  Measured
// (scripts/ubench/decode_lab.hip, profiles/r06_experiments.md): q/k/v rows 11.6 -> 10.8 us per layer; no gain without a norm (o_proj) and
// -0.3 us on the SwiGLU pair (94 VGPRs), so the launcher takes it for norm-carrying single-pass rows only (q/k/v, lm_head).
// XV = 16-B vectors of x per thread (K <= XV * 2048).  grid = ceil(N_out / 4), block 256, dynamic LDS = K * 2 bytes.
template <bool SWIGLU, bool OUT_F32, int XV>
__global__ __launch_bounds__(256) void gemv_xfirst_bf16_kernel(GemvArgs p) {
#pragma clang fp reassociate(off)                  // see gemv_bf16_kernel: the norm arithmetic is pinned to one order
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    __shared__ float red[8];
    bf16_t* xs = (bf16_t*)vl2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_out = SWIGLU ? p.N / 2 : p.N;
    const int nvec = p.K >> 3;
    // x first, then the first pass of the row's weights: both in flight before anything is waited for
    u32x4 xr[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int k = tid * 8 + i * 2048;
        if (k < p.K) xr[i] = *(const u32x4*)(p.x + k);
    }
    u32x4 wv[8], uv[8];
    auto issue_row = [&](int j, int v0) {
        const int row0 = SWIGLU ? (j >> 5) * 64 + (j & 31) : j;
        const bf16_t* w0p = p.W + (size_t)row0 * p.ldw;
        const bf16_t* w1p = w0p + (size_t)32 * p.ldw;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = v0 + i * 64 + lane;
            if (v < nvec) {
                wv[i] = __builtin_nontemporal_load((const u32x4*)(w0p + (size_t)v * 8));
                if (SWIGLU) uv[i] = __builtin_nontemporal_load((const u32x4*)(w1p + (size_t)v * 8));
            }
        }
    };
    const int j = blockIdx.x * 4 + wave;
    if (j < n_out) issue_row(j, 0);
    float rstd = 1.f;
    const bool norm = p.norm_w != nullptr || p.rms_plain;
    if (norm) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            if (tid * 8 + i * 2048 < p.K) {
                float v[8];
                unpack8(xr[i], v);
#pragma unroll
                for (int q = 0; q < 8; ++q) ss = __builtin_fmaf(v[q], v[q], ss);
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)p.K + p.eps);
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        const int k = tid * 8 + i * 2048;
        if (k < p.K) {
            u32x4 raw = xr[i];
            if (norm) {
                float v[8];
                unpack8(raw, v);
                f32x4 w0 = {1.f, 1.f, 1.f, 1.f}, w1 = w0;
                if (p.norm_w) { w0 = *(const f32x4*)(p.norm_w + k); w1 = *(const f32x4*)(p.norm_w + k + 4); }
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (v[q] * rstd) * (q < 4 ? w0[q] : w1[q - 4]);
                raw = pack8(v);
            }
            *(u32x4*)(xs + k) = raw;
        }
    }
    __syncthreads();
    if (j >= n_out) return;
    float a0 = 0.f, a1 = 0.f;
    for (int v0 = 0; v0 < nvec; v0 += 64 * 8) {
        if (v0) issue_row(j, v0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = v0 + i * 64 + lane;
            if (v < nvec) {
                const u32x4 xv = *(const u32x4*)(xs + (size_t)v * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a0 = dot2_bf16(wv[i][q], xv[q], a0);
                    if (SWIGLU) a1 = dot2_bf16(uv[i][q], xv[q], a1);
                }
            }
        }
    }
    a0 = wave_sum(a0);
    if (SWIGLU) a1 = wave_sum(a1);
    if (lane == 0) {
        float o = SWIGLU ? silu_f(a0) * a1 : a0;
        if (!SWIGLU && p.bias) o += p.bias[j];
        if (p.res) o += bf2f(p.res[j]);
        if (OUT_F32) ((float*)p.y)[j] = o;
        else ((bf16_t*)p.y)[j] = f2bf(o);
    }
}

// Multi-row form for BATCHED decode: y[b][:] = W x[b][:] for MB = 2..4 sequences in one pass over W.
// Decode is bound by streaming the weights; with MB tokens (one per sequence) sharing the stream, the weights are read once
// for MB outputs.  Same structure as gemv_bf16_kernel (one output row per wave, the row's 8 loads issued before x is
// staged); x rows live in LDS as [MB][K] bf16 (MB * K * 2 <= 64 KiB: the launcher splits larger batches), every weight
// vector is multiplied against the MB x vectors it meets.  grid = ceil(N_out / 4), block 256.
template <bool SWIGLU, bool OUT_F32, int MB, int RPW>
__global__ __launch_bounds__(256) void gemv_mr_bf16_kernel(GemvArgs p) {
#pragma clang fp reassociate(off)                  // see gemv_bf16_kernel: the norm arithmetic is pinned to one order
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    __shared__ float red[MB][4];
    bf16_t* xs = (bf16_t*)vl2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_out = SWIGLU ? p.N / 2 : p.N;
    const int nvec = p.K >> 3;
    const bool one_pass = nvec <= 512;
    u32x4 wv[8], uv[8];
    auto issue_row = [&](int j, int v0) {
        const int row0 = SWIGLU ? (j >> 5) * 64 + (j & 31) : j;
        const bf16_t* w0p = p.W + (size_t)row0 * p.ldw;
        const bf16_t* w1p = w0p + (size_t)32 * p.ldw;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = v0 + i * 64 + lane;
            if (v < nvec) {
                wv[i] = __builtin_nontemporal_load((const u32x4*)(w0p + (size_t)v * 8));
                if (SWIGLU) uv[i] = __builtin_nontemporal_load((const u32x4*)(w1p + (size_t)v * 8));
            }
        }
    };
    const int jfirst = (blockIdx.x * 4 + wave) * RPW;      // RPW consecutive output rows per wave: staging MB rows of x
    if (one_pass && jfirst < n_out) issue_row(jfirst, 0);  // (MB x the single-row prologue) is amortised over 4*RPW rows
    float rstd[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) rstd[b] = 1.f;
    if (p.norm_w) {
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            float ss = 0.f;
            for (int k = tid * 8; k < p.K; k += 2048) {
                float v[8];
                unpack8(*(const u32x4*)(p.x + (size_t)b * p.ldx + k), v);
#pragma unroll
                for (int q = 0; q < 8; ++q) ss = __builtin_fmaf(v[q], v[q], ss);
            }
            ss = wave_sum(ss);
            if (lane == 0) red[b][wave] = ss;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < MB; ++b) rstd[b] = rsqrtf(((red[b][0] + red[b][1]) + (red[b][2] + red[b][3])) / (float)p.K + p.eps);
    }
#pragma unroll
    for (int b = 0; b < MB; ++b)
        for (int k = tid * 8; k < p.K; k += 2048) {
            u32x4 raw = *(const u32x4*)(p.x + (size_t)b * p.ldx + k);
            if (p.norm_w) {
                float v[8];
                unpack8(raw, v);
                const f32x4 w0 = *(const f32x4*)(p.norm_w + k), w1 = *(const f32x4*)(p.norm_w + k + 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (v[q] * rstd[b]) * (q < 4 ? w0[q] : w1[q - 4]);
                raw = pack8(v);
            }
            *(u32x4*)(xs + (size_t)b * p.K + k) = raw;
        }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < RPW; ++r) {
    const int j = jfirst + r;
    if (j >= n_out) break;
    float a0[MB], a1[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) a0[b] = a1[b] = 0.f;
    for (int v0 = 0; v0 < nvec; v0 += 64 * 8) {
        if (!(one_pass && r == 0)) issue_row(j, v0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = v0 + i * 64 + lane;
            if (v < nvec) {
#pragma unroll
                for (int b = 0; b < MB; ++b) {
                    const u32x4 xv = *(const u32x4*)(xs + (size_t)b * p.K + (size_t)v * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a0[b] = dot2_bf16(wv[i][q], xv[q], a0[b]);
                        if (SWIGLU) a1[b] = dot2_bf16(uv[i][q], xv[q], a1[b]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) {
        a0[b] = wave_sum(a0[b]);
        if (SWIGLU) a1[b] = wave_sum(a1[b]);
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            float o = SWIGLU ? silu_f(a0[b]) * a1[b] : a0[b];
            if (!SWIGLU && p.bias) o += p.bias[j];
            if (p.res) o += bf2f(p.res[(size_t)b * p.ldres + j]);
            if (OUT_F32) ((float*)p.y)[(size_t)b * p.ldy + j] = o;
            else ((bf16_t*)p.y)[(size_t)b * p.ldy + j] = f2bf(o);
        }
    }
    }
}

// ---- the combine of the flash-decoding partials (shared by attn_decode_combine_kernel and the fused form of attn_decode_kernel)
#define COMBINE_CHUNK 256
#define COMBINE_EARLY 32
// One head's combine, 128 threads (d = 0..127 = the output dimension; the threads with d < 64 are one wave and turn the (m_i, l_i)
// pairs into weights).  ACQ = the partials were written by OTHER workgroups of the same launch (attn_decode_kernel<true>): they are
// read with agent-scope relaxed atomic loads (global_load ... sc1: bypass this CU's vector L1, the producer stored write-through).
// Every thread of the workgroup must call this the same number of times (it contains workgroup barriers).  The arithmetic -- and
// so the bits -- is the same whichever kernel runs it.
template <bool ACQ>
__device__ __forceinline__ float attn_ld(const float* p) {
    if (ACQ) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool ACQ, int TPH = 128, bool WT = false>
__device__ __forceinline__ void attn_combine_head(const float* __restrict__ src, int nsplit, int d, float* wgt, float* red,
                                                  bf16_t* outp, bool store) {
    // TPH threads work on one head (d = 0 .. TPH-1; the first 64 of them are one wave and turn the (m_i, l_i) pairs into weights);
    // thread d sums the 128 / TPH output columns d, d + TPH, ... one after the other -- every output element is the same fixed-order
    // sum whatever TPH is.  Two instantiations (plain / agent-scope loads) must give the same bits: under -ffast-math the compiler
    // is otherwise free to contract or re-associate them differently (seen on hardware: one bf16 output element in 4096 off by an
    // ulp) -> the sums are explicit FMAs in a fixed order
#pragma clang fp reassociate(off)
    // short contexts (<= COMBINE_EARLY slices = 2048 positions, TPH = 128): the column's values do not depend on the weights, so they are
    // requested BEFORE the (m, l) passes -- the whole combine is one memory round trip instead of ~four dependent ones (4.7 -> ~2.7 us
    // per layer and token).  The FMAs below run in the same order on the same values: same bits.
    constexpr bool EARLY = TPH == 128;
    float ev[EARLY ? COMBINE_EARLY : 1];
    const bool early = EARLY && nsplit <= COMBINE_EARLY;
    if constexpr (EARLY) {
        if (early) {
#pragma unroll
            for (int i = 0; i < COMBINE_EARLY; ++i) ev[i] = i < nsplit ? attn_ld<ACQ>(src + i * 130 + 2 + d) : 0.f;
        }
    }
    // ... and so are slice d's (m, l) (early: nsplit <= 32 <= TPH): the passes below then need no further memory round trip
    float pm = -1e30f, pl = 0.f;
    if (early && d < nsplit) { pm = attn_ld<ACQ>(src + d * 130); pl = attn_ld<ACQ>(src + d * 130 + 1); }
    if (d < 64) {                                             // one wave: global max M, then L = sum_i l_i 2^(m_i - M)
        float M = -1e30f;
        if (early) M = fmaxf(M, pm);
        else for (int i0 = 0; i0 < nsplit; i0 += 64) M = fmaxf(M, (i0 + d < nsplit) ? attn_ld<ACQ>(src + (i0 + d) * 130) : -1e30f);
        M = wave_max(M);
        float L = 0.f;
        if (early) { if (d < nsplit) L = __builtin_fmaf(pl, exp2f(pm - M), L); }
        else for (int i0 = 0; i0 < nsplit; i0 += 64)
            if (i0 + d < nsplit) L = __builtin_fmaf(attn_ld<ACQ>(src + (i0 + d) * 130 + 1), exp2f(attn_ld<ACQ>(src + (i0 + d) * 130) - M), L);
        L = wave_sum(L);
        if (d == 0) { red[0] = M; red[1] = 1.0f / L; }
    }
    __syncthreads();
    const float M = red[0];
    float o[128 / TPH];
#pragma unroll
    for (int c = 0; c < 128 / TPH; ++c) o[c] = 0.f;
    for (int c0 = 0; c0 < nsplit; c0 += COMBINE_CHUNK) {
        const int nc = nsplit - c0 < COMBINE_CHUNK ? nsplit - c0 : COMBINE_CHUNK;
        if (c0 > 0) __syncthreads();                          // the previous chunk's weights have been consumed
        if (early) { if (d < nc) wgt[d] = exp2f(pm - M); }
        else for (int i = d; i < nc; i += TPH) wgt[i] = exp2f(attn_ld<ACQ>(src + (c0 + i) * 130) - M);
        __syncthreads();
        if constexpr (EARLY) {
            if (early) {                                      // nc == nsplit, c0 == 0
#pragma unroll
                for (int i = 0; i < COMBINE_EARLY; ++i)
                    if (i < nc) o[0] = __builtin_fmaf(ev[i], wgt[i], o[0]);
                break;
            }
        }
#pragma unroll
        for (int c = 0; c < 128 / TPH; ++c) {
            const float* s0 = src + (size_t)c0 * 130 + 2 + d + c * TPH;
            int i = 0;
            for (; i + 8 <= nc; i += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = attn_ld<ACQ>(s0 + (i + j) * 130);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[c] = __builtin_fmaf(v[j], wgt[i + j], o[c]);
            }
            for (; i < nc; ++i) o[c] = __builtin_fmaf(attn_ld<ACQ>(s0 + i * 130), wgt[i], o[c]);
        }
    }
    if (store) {
#pragma unroll
        for (int c = 0; c < 128 / TPH; ++c) {
            const bf16_t r = f2bf(o[c] * red[1]);
            if (WT) __hip_atomic_store(outp + d + c * TPH, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through: read by other workgroups of this launch
            else outp[d + c * TPH] = r;
        }
    }
}
// ---- one 64-key slice of the decode attention, shared by attn_decode_kernel and attn_oproj_kernel (k_decode2.h): four waves (`tid` = 0..255
// inside the group of four) hold 16 keys each (kreg = the lane's K row, vv = the wave's V rows, `valid` = the lane's key is inside the
// context), the roped q heads of the block are in qs.  Scores, slice softmax (exp2 domain), P V and the {m, l, o[128]} partial of each of
// the `ng` heads -> pdst + h * hstride (only when `store`).  WT = write-through (sc1) partial stores (read by other workgroups of the same
// launch).  Contains three workgroup barriers: every thread of the workgroup calls it the same number of times.  The sums are explicit
// FMAs in a fixed order: the function is inlined into several kernels and -ffast-math would otherwise be free to contract them differently
// in each (seen on hardware: last-bit differences in the partials between two kernels sharing this code).
struct AttnSliceSmem {
    __attribute__((aligned(16))) float qs[4][128];
    __attribute__((aligned(16))) float pk[64][4];       // p[key][head]
    float wred[2][4][4];                                 // [max|sum][wave][head]
    __attribute__((aligned(16))) float oacc[4][4][128];  // [wave][head][d]
};
template <bool WT>
__device__ __forceinline__ void attn_slice_compute(AttnSliceSmem& sm, int tid, const u32x4 (&kreg)[16], const uint32_t (&vv)[16], bool valid, int ng,
                                                   float scale_log2e, float* pdst, size_t hstride, bool store) {
#pragma clang fp reassociate(off)
    constexpr int HD = 128;
    const int lane = tid & 63, wave = tid >> 6;
    const int kl = lane & 15, hh = lane >> 4;
    const int hq = hh < ng ? hh : ng - 1;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float kv[8];
        unpack8(kreg[c], kv);
        const f32x4 q0 = *(const f32x4*)&sm.qs[hq][c * 8], q1 = *(const f32x4*)&sm.qs[hq][c * 8 + 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(kv[4 + j], q1[j], __builtin_fmaf(kv[j], q0[j], acc));
    }
    const float s = valid ? acc * scale_log2e : -1e30f;
    float mx = s;
#pragma unroll
    for (int msk = 8; msk >= 1; msk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, msk));     // over the 16 keys of (wave, head)
    if (kl == 0) sm.wred[0][wave][hh] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(sm.wred[0][0][hh], sm.wred[0][1][hh]), fmaxf(sm.wred[0][2][hh], sm.wred[0][3][hh]));
    const float pv = valid ? exp2f(s - m) : 0.f;
    float sum = pv;
#pragma unroll
    for (int msk = 8; msk >= 1; msk >>= 1) sum += __shfl_xor(sum, msk);
    if (kl == 0) sm.wred[1][wave][hh] = sum;
    sm.pk[wave * 16 + kl][hh] = pv;
    __syncthreads();

    float o[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h) o[h][0] = o[h][1] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x4 p4 = *(const f32x4*)sm.pk[wave * 16 + i];
        const float v0 = e_lo(vv[i]), v1 = e_hi(vv[i]);
#pragma unroll
        for (int h = 0; h < 4; ++h) { o[h][0] = __builtin_fmaf(p4[h], v0, o[h][0]); o[h][1] = __builtin_fmaf(p4[h], v1, o[h][1]); }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) { sm.oacc[wave][h][lane * 2] = o[h][0]; sm.oacc[wave][h][lane * 2 + 1] = o[h][1]; }
    __syncthreads();
    // a thread stores a PAIR of floats (8-byte aligned: slices are 130 floats apart): half as many stores, and a write-through store is one
    // fabric write whatever its width
    auto put2 = [](float* p, float v0, float v1) {
        if (WT) {
            const f32x2 v = {v0, v1};
            __hip_atomic_store((uint64_t*)p, __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through (sc1)
        } else {
            *(f32x2*)p = f32x2{v0, v1};
        }
    };
    if (store) {
        for (int t = tid; t < ng * (HD / 2); t += 256) {
            const int h = t / (HD / 2), d = (t % (HD / 2)) * 2;
            float* dst = pdst + (size_t)h * hstride;
            put2(dst + 2 + d, ((sm.oacc[0][h][d] + sm.oacc[1][h][d]) + sm.oacc[2][h][d]) + sm.oacc[3][h][d],
                 ((sm.oacc[0][h][d + 1] + sm.oacc[1][h][d + 1]) + sm.oacc[2][h][d + 1]) + sm.oacc[3][h][d + 1]);
            if (d == 0)
                put2(dst, fmaxf(fmaxf(sm.wred[0][0][h], sm.wred[0][1][h]), fmaxf(sm.wred[0][2][h], sm.wred[0][3][h])),
                     ((sm.wred[1][0][h] + sm.wred[1][1][h]) + sm.wred[1][2][h]) + sm.wred[1][3][h]);
        }
    }
}

// ---- decode attention (flash-decoding) fused with RoPE and the KV-cache append of the new token.
// qkv [(nh+2*nkv)*128] = the un-roped fused projection of the ONE new token at position pos (pos = *pos_dev when pos_dev
// is non-null, so a captured hipGraph replays with a moving position).  grid = (nsplit_cap, nkv, ceil(group/4)), 256
// threads: one workgroup per (64-key slice, kv head, block of <= 4 q heads of the group), 4 waves x 16 keys; slices at or
// beyond ctx = pos+1 exit at once.  (Mistral-7B: group 4 = one block; Qwen2-7B: group 7 = two blocks.)
// A lane is (key = lane&15, head-in-block = lane>>4): a K row is read once and scored against the block's q heads;
// V rows are requested up front (they do not depend on the scores) so the kernel is ONE memory round trip.
// The workgroup(s) whose slice contains pos rope k_new and append k_new / v_new to the cache (DynamicCache.update) first
// (with two head blocks both write the same bytes, and each reads the row back only after its own barrier).
// partial: fp32 [nh][nsplit_cap][130] = {m (exp2 domain), l, o[128]}
// FUSED (single sequence, the stage-level decode step): the combine runs inside this launch -- every workgroup stores its partials
// write-through (agent-scope relaxed atomic stores = `global_store ... sc1`), drains them, and takes a ticket on its kv head's
// counter `cnt[hk]` (zeroed by the caller before the launch); the workgroup that draws the last ticket of the head combines the
// `group` q heads of that kv head (attn_combine_head<true>: the same arithmetic as attn_decode_combine_kernel, partials read past
// the vector L1) and writes `out`.  One launch and one kernel boundary less per layer and token than attn + combine
// (7.6 + 4.7 us + a boundary -> measured in profiles/r03_experiments.md).  Round 1 tried the same election with an agent-scope
// RELEASE fence in every slice (buffer_wbl2) and lost; write-through stores need no fence (guide G16 R1).
template <bool FUSED>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kcache,
                                                          bf16_t* __restrict__ vcache, const float* __restrict__ cos_t,
                                                          const float* __restrict__ sin_t, float* __restrict__ partial,
                                                          int nh, int group, int nkv, int smax, int pos_arg,
                                                          const int* __restrict__ pos_dev, float scale_log2e,
                                                          long qkv_bs, long cache_bs, long partial_bs,
                                                          int* __restrict__ cnt, bf16_t* __restrict__ out) {
    // batched decode: blockIdx.y = kv head + nkv * sequence; sequence b uses qkv + b*qkv_bs, caches + b*cache_bs,
    // partial + b*partial_bs and position pos_dev[b] (single sequence: strides 0, b = 0)
    constexpr int HD = 128, HALF = 64;
    __shared__ AttnSliceSmem sm;
    auto& qs = sm.qs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kl = lane & 15;
    const int split = blockIdx.x, hk = (int)blockIdx.y % nkv, bseq = (int)blockIdx.y / nkv, nsplit = gridDim.x;
    const int h0 = blockIdx.z * 4;                                  // first q head (within the group) of this block
    const int ng = group - h0 < 4 ? group - h0 : 4;                 // q heads of this block
    const int pos = pos_dev ? pos_dev[bseq] : pos_arg;
    qkv += (size_t)bseq * qkv_bs;
    kcache += (size_t)bseq * cache_bs;
    vcache += (size_t)bseq * cache_bs;
    partial += (size_t)bseq * partial_bs;
    // pos >= smax can only come from a device-side position that ran past the cache (hipGraph replay at the cache end): such a
    // step must not touch cos/sin row `pos` or cache row `pos`; the host discards its result (decoder.generate `last` step)
    if (pos >= smax) return;
    const int ctx = pos + 1;
    const int k0 = split * 64;
    if (k0 >= ctx) return;
    bf16_t* Kb = kcache + (size_t)hk * smax * HD;
    bf16_t* Vb = vcache + (size_t)hk * smax * HD;
    const float* cp = cos_t + (size_t)pos * HALF;
    const float* sp = sin_t + (size_t)pos * HALF;

    // roped q of the `group` heads of this kv head -> LDS (rounded through bf16 like the unfused path)
    for (int t = tid; t < ng * HALF; t += 256) {
        const int h = t / HALF, d = t % HALF;
        const bf16_t* qh = qkv + (size_t)(hk * group + h0 + h) * HD;
        float o1, o2;
        rope_pair(bf2f(qh[d]), bf2f(qh[d + HALF]), cp[d], sp[d], o1, o2);
        qs[h][d] = bf2f(f2bf(o1));                      // rounded through bf16 like the prefill's roped q
        qs[h][d + HALF] = bf2f(f2bf(o2));
    }
    // the slice that owns the new position appends roped k_new and v_new to the cache before anyone reads row `pos`
    if (pos >= k0 && pos < k0 + 64 && tid < HALF) {
        const bf16_t* kn = qkv + (size_t)(nh + hk) * HD;
        const bf16_t* vn = qkv + (size_t)(nh + nkv + hk) * HD;
        float o1, o2;
        rope_pair(bf2f(kn[tid]), bf2f(kn[tid + HALF]), cp[tid], sp[tid], o1, o2);
        Kb[(size_t)pos * HD + tid] = f2bf(o1);
        Kb[(size_t)pos * HD + tid + HALF] = f2bf(o2);
        Vb[(size_t)pos * HD + tid] = vn[tid];
        Vb[(size_t)pos * HD + tid + HALF] = vn[tid + HALF];
    }
    __syncthreads();       // workgroup-scope release/acquire: the appended rows are visible to this workgroup's loads

    // V rows of this wave's 16 keys: lane owns dims 2*lane, 2*lane+1 (256 B per key per wave, coalesced)
    uint32_t vv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int key = k0 + wave * 16 + i;
        key = key < ctx ? key : ctx - 1;
        vv[i] = *(const uint32_t*)(Vb + (size_t)key * HD + lane * 2);
    }
    const int key = k0 + wave * 16 + kl;
    const bool valid = key < ctx;
    const bf16_t* kr = Kb + (size_t)(valid ? key : ctx - 1) * HD;
    u32x4 kreg[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) kreg[c] = *(const u32x4*)(kr + c * 8);

    attn_slice_compute<FUSED>(sm, tid, kreg, vv, valid, ng, scale_log2e,
                              partial + ((size_t)(hk * group + h0) * nsplit + split) * 130, (size_t)nsplit * 130, true);
    if constexpr (FUSED) {
        __shared__ float wgtf[4][COMBINE_CHUNK];
        __shared__ float redf[4][2];
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores (guide G16 R1)
        __syncthreads();
        const int nlive = (ctx + 63) >> 6;                          // slices that reach this point, per head block
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(cnt + hk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = ticket == nlive * (int)gridDim.z - 1;
        }
        __syncthreads();
        if (!s_last) return;
        // the elected workgroup: four heads at a time, one wave each.  The partial buffer is reused by every layer and token, so this
        // CU's vector L1 may hold stale lines of it: the combine reads with agent-scope atomic loads (sc1: past the L1; the producers
        // stored write-through), which is the "sc1 both sides" form of guide G16 -- no fence.
        const int q4 = tid >> 6, d = tid & 63;
        for (int hp = 0; hp < group; hp += 4) {
            const bool act = hp + q4 < group;
            const int head = hk * group + (act ? hp + q4 : group - 1);
            attn_combine_head<true, 64>(partial + (size_t)head * nsplit * 130, nlive < nsplit ? nlive : nsplit, d, wgtf[q4], redf[q4],
                                        out + (size_t)head * HD, act);
            __syncthreads();                                        // the weights / (M, 1/L) slots are reused by the next four
        }
    }
}

// grid = nh, block 128: out[h*128 + d] = sum_i o_i[d] 2^(m_i - M) / sum_i l_i 2^(m_i - M) over the ceil(ctx/64) live slices.
// Wave 0 turns the (m_i, l_i) pairs into weights (64 slices per pass, any slice count: running max M over the passes first,
// then the weights against the final M); then every thread sums its column with independent, coalesced loads (no dependent
// chain over the slices).  The weights live in LDS in chunks of COMBINE_CHUNK slices, so the context length is unbounded.
__global__ __launch_bounds__(128) void attn_decode_combine_kernel(const float* __restrict__ partial, bf16_t* __restrict__ out,
                                                                  int nsplit_cap, int pos_arg, const int* __restrict__ pos_dev,
                                                                  long partial_bs, long out_bs) {
    __shared__ float wgt[COMBINE_CHUNK];
    __shared__ float red[2];                                  // {M, 1/L}
    const int h = blockIdx.x, d = threadIdx.x, bseq = blockIdx.y;       // grid = (nh, sequences)
    const int pos = pos_dev ? pos_dev[bseq] : pos_arg;
    int nsplit = (pos + 64) >> 6;
    // a position at or beyond the launch's capacity (exhausted cache under hipGraph replay) would index past `partial`:
    // clamp to the slices the launch produced -- the host never uses the result of such a step (decoder.generate)
    nsplit = nsplit < nsplit_cap ? nsplit : nsplit_cap;
    const float* src = partial + (size_t)bseq * partial_bs + (size_t)h * nsplit_cap * 130;
    attn_combine_head<false>(src, nsplit, d, wgt, red, out + (size_t)bseq * out_bs + h * 128, true);
}

// first index of the maximum of logits[V] (fp32) -> *tok (int32) and hist[step]; one block of 1024 threads.
// state != null (hipGraph-replayable decode): step = state[1]; afterwards state[0] (position) and state[1] advance by one.
// zero / nzero: int32 words this launch clears first (the ticket counters of the fused attention launches of the step that follows:
// a kernel-side clear instead of a memset node, which replayed wrongly from a captured hipGraph for a 32-byte range on ROCm 7.2).
// embed != null (the stage-level decode step): the block then copies row `tok` of the embedding table (D bf16) to x0 -- the
// embed_rows launch that used to follow (every launch of the decode graph costs ~4.5 us whatever it does).
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int V, int* __restrict__ tok,
                                                      int* __restrict__ hist, int step, int* __restrict__ state,
                                                      int* __restrict__ zero, int nzero, const bf16_t* __restrict__ embed,
                                                      bf16_t* __restrict__ x0, int D) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int s_tok;
    for (int i = threadIdx.x; i < nzero; i += 1024) zero[i] = 0;
    float best = -3.4e38f;
    int idx = 0x7fffffff;
    // eight independent loads per trip (the scalar loop was one L2 round trip per element: ~31 dependent trips for V = 32000, 12 us);
    // the compares stay in increasing index order, so ties still resolve to the first index
    for (int i0 = threadIdx.x; i0 < V; i0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i0 + k * 1024 < V) ? logits[i0 + k * 1024] : -3.4e38f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (v[k] > best) { best = v[k]; idx = i0 + k * 1024; }
    }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
        const float ov = __shfl_xor(best, msk);
        const int oi = __shfl_xor(idx, msk);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        *tok = idx;
        s_tok = idx;
        if (state) {
            if (hist) hist[state[1]] = idx;
            state[0] += 1;
            state[1] += 1;
        } else if (hist) {
            hist[step] = idx;
        }
    }
    if (embed) {
        __syncthreads();
        const bf16_t* src = embed + (size_t)s_tok * D;
        for (int c = threadIdx.x * 8; c < D; c += 8 * 1024) *(u32x4*)(x0 + c) = *(const u32x4*)(src + c);
    }
}

// out[i, :] = table[ids[i], :] (bf16 rows of D elements); ids int32 on the device; grid = n, block 128
__global__ __launch_bounds__(128) void embed_rows_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ table,
                                                         bf16_t* __restrict__ out, int D, int ldo) {
    const int i = blockIdx.x;
    const bf16_t* src = table + (size_t)ids[i] * D;
    for (int c = threadIdx.x * 8; c < D; c += 1024) *(u32x4*)(out + (size_t)i * ldo + c) = *(const u32x4*)(src + c);
}

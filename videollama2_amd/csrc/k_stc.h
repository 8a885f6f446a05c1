// STC connector direct kernels (videollama2/model/projector.py:133-215; timm 1.0.3 regnet.Bottleneck / SEModule /
// LayerNormAct2d as restated in oracle/shims/timm).  Activations are channels-last ("token-major": [F, H, W, C]) so
// the 1x1 convs are plain GEMMs (k_gemm.h) and LayerNorm2d is a row LayerNorm.  All kernels here are HBM/L2-bound.
//   dwconv_ln_silu_kernel : depthwise 3x3 (pad 1, groups=C, no bias) + LayerNorm2d(eps 1e-5) + SiLU, one workgroup
//                           per output position (the whole C row lives in the workgroup, so the LN is fused).
//   chan_mean_kernel      : SE squeeze: mean over the H*W positions of each frame -> fp32 [F, C].
//   small_linear_kernel   : SE excite: the two tiny 1x1 convs on [F, C] vectors (+bias, SiLU / sigmoid), fp32 in/out.
//   se_scale_kernel       : x *= gate[f, c].
#pragma once
#include "dev_common.h"

__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 8 floats, LDS */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();                       // protect `red` against the previous use
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// x, y: [F, H, W, C] bf16; wt: [9][C] fp32 tap-major (tap = ky*3+kx); grid = F*H*W, block 256; NVT*256*8 >= C
template <int NVT>
__global__ __launch_bounds__(256) void dwconv_ln_silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                             const float* __restrict__ wt, const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, int H, int W, int C, float eps) {
    __shared__ float red[8];
    const int pos = blockIdx.x;
    const int w0 = pos % W, h0 = (pos / W) % H;
    const size_t fbase = (size_t)(pos - (h0 * W + w0)) * C;      // frame base (elements)
    float acc[NVT][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        if (c < C) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int hh = h0 + ky - 1, ww = w0 + kx - 1;
                    if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                        float xv[8];
                        unpack8(*(const u32x4*)(x + fbase + (size_t)(hh * W + ww) * C + c), xv);
                        const float* wp = wt + (ky * 3 + kx) * C + c;
                        const f32x4 wa = *(const f32x4*)wp, wb = *(const f32x4*)(wp + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { acc[i][j] += xv[j] * wa[j]; acc[i][4 + j] += xv[4 + j] * wb[j]; }
                    }
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j];
        }
    }
    const float mean = block_sum_256(s, red) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = acc[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(block_sum_256(q, red) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < C) {
            const f32x4 g0 = *(const f32x4*)(lnw + c), g1 = *(const f32x4*)(lnw + c + 4);
            const f32x4 b0 = *(const f32x4*)(lnb + c), b1 = *(const f32x4*)(lnb + c + 4);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = silu_f((acc[i][j] - mean) * rstd * (j < 4 ? g0[j] : g1[j - 4]) + (j < 4 ? b0[j] : b1[j - 4]));
            *(u32x4*)(y + (size_t)pos * C + c) = pack8(o);
        }
    }
}

// x [F, HW, C] bf16 -> mean [F, C] fp32; grid = (C/64, F), block 256 = 8 channel-vectors x 32 position lanes
__global__ __launch_bounds__(256) void chan_mean_kernel(const bf16_t* __restrict__ x, float* __restrict__ mean, int HW, int C) {
    __shared__ float part[32][65];
    const int cv = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = blockIdx.x * 64 + cv * 8, f = blockIdx.y;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    for (int p = pl; p < HW; p += 32) {
        float v[8];
        unpack8(*(const u32x4*)(x + ((size_t)f * HW + p) * C + c), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[pl][cv * 8 + j] = a[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += part[i][threadIdx.x];
        mean[(size_t)f * C + blockIdx.x * 64 + threadIdx.x] = t / (float)HW;
    }
}

// out[f][n] = act( sum_k W[n][k] * x[f][k] + b[n] );  x fp32 [F,K], W bf16 [N,K], out fp32 [F,N]; one wave per n;
// act: 0 none, 1 SiLU, 2 sigmoid.  grid = ceil(N/4), block 256.  F is processed 8 frames at a time.
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ b, float* __restrict__ out, int F,
                                                           int N, int K, int act) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int f0 = 0; f0 < F; f0 += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            float w[8];
            unpack8(*(const u32x4*)(W + (size_t)n * K + k), w);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (f0 + i < F) {
                    const float* xp = x + (size_t)(f0 + i) * K + k;
                    const f32x4 xa = *(const f32x4*)xp, xb = *(const f32x4*)(xp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i] += w[j] * xa[j] + w[4 + j] * xb[j];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = wave_sum(acc[i]);
            if (lane == 0 && f0 + i < F) {
                float v = t + (b ? b[n] : 0.f);
                if (act == 1) v = silu_f(v);
                if (act == 2) v = sigmoid_f(v);
                out[(size_t)(f0 + i) * N + n] = v;
            }
        }
    }
}

// x [F, HW, C] bf16 (in place) *= gate [F, C] fp32; grid-stride over 8-element vectors
__global__ __launch_bounds__(256) void se_scale_kernel(bf16_t* __restrict__ x, const float* __restrict__ gate, int HW, int C,
                                                       size_t nvec) {
    const int cvec = C >> 3;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
        const size_t row = v / cvec;
        const int c = (int)(v - row * cvec) * 8;
        const size_t f = row / HW;
        float a[8];
        unpack8(*(const u32x4*)(x + v * 8), a);
        const float* g = gate + f * C + c;
        const f32x4 g0 = *(const f32x4*)g, g1 = *(const f32x4*)(g + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] *= g0[j]; a[4 + j] *= g1[j]; }
        *(u32x4*)(x + v * 8) = pack8(a);
    }
}

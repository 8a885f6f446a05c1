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

// The same operator with a workgroup owning DW_P = 4 consecutive output positions of one image row: per input row it loads the
// 6-position window once (packed bf16 in registers) and each tap's weights once for the 4 outputs -- 73 KB through L2 per
// output instead of 9 x C x 6 B = 221 KB.  Per output the taps are accumulated in the same (ky, kx) order.  Used for wide
// rows (the launcher takes it from W >= 16: the 24x24 grid of stage s1, 97 -> 74 us at 16 frames); on the 13x13 grid of
// stage s2 a quarter of the workgroups would hold one live position.  grid = F * H * ceil(W / DW_P).
#define DW_P 4
// (NVT <= 2 is held to 128 VGPRs = four waves per SIMD: 130 -> 128 registers with 12 B/lane of scratch outside the tap loop,
// 72.3 -> 69.0 us on the s1 grid, scripts/stc_bench.py)
template <int NVT>
__global__ __launch_bounds__(256, NVT <= 2 ? 4 : 2) void dwconv4_ln_silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                              const float* __restrict__ wt, const float* __restrict__ lnw,
                                                              const float* __restrict__ lnb, int H, int W, int C, float eps) {
#pragma clang fp reassociate(off)
    __shared__ float red[DW_P][4];
    const int gpr = (W + DW_P - 1) / DW_P;                        // position groups per image row
    const int grp = blockIdx.x % gpr, rowid = blockIdx.x / gpr;    // rowid = f * H + h0
    const int h0 = rowid % H, w0 = grp * DW_P;
    const size_t fbase = (size_t)(rowid - h0) * W * C;             // frame base (elements)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[DW_P][NVT][8];
#pragma unroll
    for (int p = 0; p < DW_P; ++p)
#pragma unroll
        for (int i = 0; i < NVT; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][i][j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hh = h0 + ky - 1;
        if (hh < 0 || hh >= H) continue;                           // zero padding: the whole input row contributes nothing
        u32x4 win[DW_P + 2][NVT];
#pragma unroll
        for (int q = 0; q < DW_P + 2; ++q) {
            const int ww = w0 + q - 1;
#pragma unroll
            for (int i = 0; i < NVT; ++i) {
                const int c = (i * 256 + threadIdx.x) * 8;
                const u32x4 z = {0u, 0u, 0u, 0u};
                win[q][i] = (ww >= 0 && ww < W && c < C) ? *(const u32x4*)(x + fbase + (size_t)(hh * W + ww) * C + c) : z;
            }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int i = 0; i < NVT; ++i) {
                const int c = (i * 256 + threadIdx.x) * 8;
                if (c >= C) continue;
                const float* wp = wt + (ky * 3 + kx) * C + c;
                const f32x4 wa = *(const f32x4*)wp, wb = *(const f32x4*)(wp + 4);
#pragma unroll
                for (int p = 0; p < DW_P; ++p) {
                    const int ww = w0 + p + kx - 1;
                    if (ww < 0 || ww >= W) continue;               // padded tap: skipped, as in the one-position kernel
                    float xv[8];
                    unpack8(win[p + kx][i], xv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[p][i][j] = __builtin_fmaf(xv[j], wa[j], acc[p][i][j]);
                        acc[p][i][4 + j] = __builtin_fmaf(xv[4 + j], wb[j], acc[p][i][4 + j]);
                    }
                }
            }
    }
    // LayerNorm statistics of the DW_P rows: one LDS round per statistic for all of them
    auto block_sums = [&](float (&v)[DW_P]) {
#pragma unroll
        for (int p = 0; p < DW_P; ++p) v[p] = wave_sum(v[p]);
        __syncthreads();                                           // protect `red` against the previous use
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < DW_P; ++p) red[p][wave] = v[p];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < DW_P; ++p) v[p] = (red[p][0] + red[p][1]) + (red[p][2] + red[p][3]);
    };
    float st[DW_P];
#pragma unroll
    for (int p = 0; p < DW_P; ++p) {
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NVT; ++i)
            if ((i * 256 + (int)threadIdx.x) * 8 < C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sm += acc[p][i][j];
            }
        st[p] = sm;
    }
    block_sums(st);
    float mean[DW_P];
#pragma unroll
    for (int p = 0; p < DW_P; ++p) {
        mean[p] = st[p] / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NVT; ++i)
            if ((i * 256 + (int)threadIdx.x) * 8 < C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = acc[p][i][j] - mean[p]; q = __builtin_fmaf(d, d, q); }
            }
        st[p] = q;
    }
    block_sums(st);
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c >= C) continue;
        const f32x4 g0 = *(const f32x4*)(lnw + c), g1 = *(const f32x4*)(lnw + c + 4);
        const f32x4 b0 = *(const f32x4*)(lnb + c), b1 = *(const f32x4*)(lnb + c + 4);
#pragma unroll
        for (int p = 0; p < DW_P; ++p) {
            if (w0 + p >= W) continue;
            const float rstd = rsqrtf(st[p] / (float)C + eps);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = silu_f(((acc[p][i][j] - mean[p]) * rstd) * (j < 4 ? g0[j] : g1[j - 4]) + (j < 4 ? b0[j] : b1[j - 4]));
            *(u32x4*)(y + ((size_t)rowid * W + w0 + p) * C + c) = pack8(o);
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

// out[f][n] = act( sum_k W[n][k] * x[f][k] + b[n] );  x fp32 [F,K], W bf16 [N,K], out fp32 [F,N]; act: 0 none, 1 SiLU,
// 2 sigmoid.  The cost is re-reading x, not the weights (one-output-per-wave forms pull F*K*4 bytes of x per output
// from L2: 256 MB for the 4096 -> 1024 SE squeeze, 29-33 us).  Here a workgroup owns SL_NB = 8 outputs and 8 frames at a
// time: each lane keeps its k-slice of the 8 frames in registers (lane owns 8 consecutive k per 2048-wide step, up to two
// steps per 4096-wide chunk) and streams the 8 weight rows past it; partial sums meet in LDS.  The k -> lane assignment
// does not depend on F, so a frame's result is the same whichever rank computes it.
#define SL_NB 8
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ b, float* __restrict__ out, int F,
                                                           int N, int K, int act) {
    // -ffast-math may re-associate the unrolled FMA chains differently per frame slot; a frame's result must not depend on
    // which slot (i.e. how many frames this rank holds) it is computed in -> fixed association inside this kernel
#pragma clang fp reassociate(off)
    __shared__ float part[4][SL_NB][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * SL_NB;
    for (int f0 = (int)blockIdx.y * 8; f0 < F; f0 += 8 * (int)gridDim.y) {      // grid.y = groups of 8 frames (a frame's bits do not depend on it)
        float acc[SL_NB][8];
#pragma unroll
        for (int j = 0; j < SL_NB; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
        for (int kc = 0; kc < K; kc += 4096) {
            f32x4 xa[2][8], xb[2][8];
            int kk[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int k = kc + st * 2048 + threadIdx.x * 8;
                kk[st] = k < K ? k : -1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = f0 + i < F ? f0 + i : F - 1;
                    const float* xp = x + (size_t)f * K + (k < K ? k : 0);
                    xa[st][i] = *(const f32x4*)xp;
                    xb[st][i] = *(const f32x4*)(xp + 4);
                }
            }
#pragma unroll
            for (int j = 0; j < SL_NB; ++j) {
                const int n = n0 + j < N ? n0 + j : N - 1;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    if (kk[st] >= 0) {
                        float w[8];
                        unpack8(*(const u32x4*)(W + (size_t)n * K + kk[st]), w);
#pragma unroll
                        for (int i = 0; i < 8; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[j][i] = __builtin_fmaf(w[4 + e], xb[st][i][e], __builtin_fmaf(w[e], xa[st][i][e], acc[j][i]));
                    }
                }
            }
        }
        // 64 partial sums per lane -> lane L ends up with the wave total of value L (= j*8 + i): a reduce-scatter butterfly,
        // 63 shuffles instead of 64 full reductions (384)
        float* v = &acc[0][0];
#pragma unroll
        for (int m = 32, half = 32; m >= 1; m >>= 1, half >>= 1) {
            const bool up = lane & m;
#pragma unroll
            for (int k = 0; k < half; ++k) {
                const float send = up ? v[k] : v[k + half];
                const float keep = up ? v[k + half] : v[k];
                v[k] = keep + __shfl_xor(send, m);
            }
        }
        part[wave][lane >> 3][lane & 7] = v[0];
        __syncthreads();
        if (threadIdx.x < SL_NB * 8) {
            const int j = threadIdx.x >> 3, i = threadIdx.x & 7, n = n0 + j;
            if (n < N && f0 + i < F) {
                float v = part[0][j][i] + part[1][j][i] + part[2][j][i] + part[3][j][i] + (b ? b[n] : 0.f);
                if (act == 1) v = silu_f(v);
                if (act == 2) v = sigmoid_f(v);
                out[(size_t)(f0 + i) * N + n] = v;
            }
        }
        __syncthreads();
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

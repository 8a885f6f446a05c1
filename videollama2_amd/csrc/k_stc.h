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

// ---- strip form (round 4).  What bounded the two kernels above was L2, not HBM: per output position they pull 9 x C fp32 taps (147 KB at C = 4096)
// next to 4.5 - 9 activation rows, ~10 TB/s of L2 traffic for 2.2 TB/s of HBM.  Here the taps are staged ONCE per workgroup into LDS (packed to the
// element type they came from: 9 x C x 2 B = 72 KB) and the workgroups are persistent: 512 threads = two TEAMS of four waves, two workgroups per CU
// (16 waves, 144 KB of LDS), each team walking units of DWS_P = 3 consecutive positions of one image row (a 24-wide row = 8 units; 16 frames = 3072
// units = exactly 3 per team of a 256-CU launch).  Teams are frame-aligned: team g serves frame g / TPF and the units j, j + TPF, ... of it, so the SE
// squeeze (timm SEModule: x.mean((2, 3)), projector.py:133 via regnet Bottleneck) falls out as ONE [C] partial sum per team (`psum`, of the ROUNDED
// outputs, as chan_mean_kernel reads them) that chan_psum_finish_kernel adds up in slot order -- deterministic, and a frame's bits do not depend on
// how many frames the launch holds as long as TPF is the same (the launcher derives TPF from the grid, not from F: see vl2_dwconv3x3_ln_silu_mean).
// Per output the taps accumulate in the same (ky, kx) order with the same fmaf chain as dwconv4_ln_silu_kernel, so the two agree bit for bit.
#define DWS_P 3
template <int NVT>
__global__ __launch_bounds__(512, NVT <= 2 ? 4 : 2) void dwconv_strip_ln_silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ wt,
                                                                      const float* __restrict__ lnw, const float* __restrict__ lnb, float* __restrict__ psum,
                                                                      int F, int H, int W, int C, float eps, int TPF, int iters) {
#pragma clang fp reassociate(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    u32x4* wl = (u32x4*)vl2_smem;                                 // [9][C/8] packed taps
    __shared__ float red[2][DWS_P][4];
    const int wv8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);     // wave-uniform: the team's unit walk stays in scalar registers
    const int team = wv8 >> 2, wave = wv8 & 3, tt = threadIdx.x & 255, lane = tt & 63;
    const int cv = C >> 3;
    for (int v = threadIdx.x; v < 9 * cv; v += 512) {
        const float* wp = wt + (size_t)v * 8;
        const f32x4 a = *(const f32x4*)wp, b = *(const f32x4*)(wp + 4);
        const float t[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        wl[v] = pack8(t);
    }
    __syncthreads();
    const int g = blockIdx.x * 2 + team, f = g / TPF, j = g - f * TPF;
    const bool live = f < F;
    const int upr = (W + DWS_P - 1) / DWS_P, U = H * upr;         // units per image row / per frame
    const size_t fbase = (size_t)(live ? f : 0) * H * W * C;
    float cs[NVT][8];
#pragma unroll
    for (int i = 0; i < NVT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[i][e] = 0.f;
    auto team_sums = [&](float (&v)[DWS_P]) {
#pragma unroll
        for (int p = 0; p < DWS_P; ++p) v[p] = wave_sum(v[p]);
        __syncthreads();                                           // protect `red` against the previous use (both teams run the same sequence)
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < DWS_P; ++p) red[team][p][wave] = v[p];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < DWS_P; ++p) v[p] = (red[team][p][0] + red[team][p][1]) + (red[team][p][2] + red[team][p][3]);
    };
    for (int it = 0; it < iters; ++it) {
        const int u = j + it * TPF;
        const bool valid = live && u < U;
        const int h0 = valid ? u / upr : 0, w0 = valid ? (u - h0 * upr) * DWS_P : 0;
        float acc[DWS_P][NVT][8];
#pragma unroll
        for (int p = 0; p < DWS_P; ++p)
#pragma unroll
            for (int i = 0; i < NVT; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[p][i][e] = 0.f;
        if (valid) {
            // channel slab outermost: 5 window vectors live at a time (the 4-position form keeps 12; with the team sums riding along that spills)
#pragma unroll
            for (int i = 0; i < NVT; ++i) {
                const int c = (i * 256 + tt) * 8;
                if (c >= C) continue;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int hh = h0 + ky - 1;
                    if (hh < 0 || hh >= H) continue;               // zero padding: the whole input row contributes nothing
                    u32x4 win[DWS_P + 2];
#pragma unroll
                    for (int q = 0; q < DWS_P + 2; ++q) {
                        const int ww = w0 + q - 1;
                        const u32x4 z = {0u, 0u, 0u, 0u};                // a padded column: a zero vector, fmaf(0, w, acc) == acc (no branch per tap)
                        const u32x4 ld = *(const u32x4*)(x + fbase + (size_t)(hh * W + (ww < 0 ? 0 : ww >= W ? W - 1 : ww)) * C + c);
                        win[q] = (ww >= 0 && ww < W) ? ld : z;
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float wv[8];
                        unpack8(wl[(ky * 3 + kx) * cv + i * 256 + tt], wv);
#pragma unroll
                        for (int p = 0; p < DWS_P; ++p) {
                            float xv[8];
                            unpack8(win[p + kx], xv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[p][i][e] = __builtin_fmaf(xv[e], wv[e], acc[p][i][e]);
                        }
                    }
                }
            }
        }
        float st[DWS_P];
#pragma unroll
        for (int p = 0; p < DWS_P; ++p) {
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < NVT; ++i)
                if ((i * 256 + tt) * 8 < C) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sm += acc[p][i][e];
                }
            st[p] = sm;
        }
        team_sums(st);
        float mean[DWS_P];
#pragma unroll
        for (int p = 0; p < DWS_P; ++p) {
            mean[p] = st[p] / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NVT; ++i)
                if ((i * 256 + tt) * 8 < C) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = acc[p][i][e] - mean[p]; q = __builtin_fmaf(d, d, q); }
                }
            st[p] = q;
        }
        team_sums(st);
        if (!valid) continue;
#pragma unroll
        for (int i = 0; i < NVT; ++i) {
            const int c = (i * 256 + tt) * 8;
            if (c >= C) continue;
            const f32x4 g0 = *(const f32x4*)(lnw + c), g1 = *(const f32x4*)(lnw + c + 4);
            const f32x4 b0 = *(const f32x4*)(lnb + c), b1 = *(const f32x4*)(lnb + c + 4);
#pragma unroll
            for (int p = 0; p < DWS_P; ++p) {
                if (w0 + p >= W) continue;
                const float rstd = rsqrtf(st[p] / (float)C + eps);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = silu_f(((acc[p][i][e] - mean[p]) * rstd) * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]));
                const u32x4 pk = pack8(o);
                *(u32x4*)(y + fbase + ((size_t)h0 * W + w0 + p) * C + c) = pk;
                unpack8(pk, o);                                    // the squeeze sums what the tensor holds
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[i][e] += o[e];
            }
        }
    }
    if (psum && live) {
#pragma unroll
        for (int i = 0; i < NVT; ++i) {
            const int c = (i * 256 + tt) * 8;
            if (c >= C) continue;
            float* dst = psum + (size_t)g * C + c;
            *(f32x4*)dst = f32x4{cs[i][0], cs[i][1], cs[i][2], cs[i][3]};
            *(f32x4*)(dst + 4) = f32x4{cs[i][4], cs[i][5], cs[i][6], cs[i][7]};
        }
    }
}

// psum [F][NP][C] fp32 (one row per team of dwconv_strip_ln_silu_kernel) -> mean [F][C] = (sum over the NP rows, in slot order) * inv_hw.
// grid = (C/256, F), block 256 = 64 channel quads x 4 slot phases.
__global__ __launch_bounds__(256) void chan_psum_finish_kernel(const float* __restrict__ psum, float* __restrict__ mean, int NP, int C, float inv_hw) {
#pragma clang fp reassociate(off)
    __shared__ f32x4 part[4][64];
    const int q = threadIdx.x & 63, ph = threadIdx.x >> 6, f = blockIdx.y;
    const int c = blockIdx.x * 256 + q * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int s = ph; s < NP; s += 4) {
            const f32x4 v = *(const f32x4*)(psum + ((size_t)f * NP + s) * C + c);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
    part[ph][q] = a;
    __syncthreads();
    if (ph == 0 && c < C) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = ((part[0][q][e] + part[1][q][e]) + (part[2][q][e] + part[3][q][e])) * inv_hw;
        *(f32x4*)(mean + (size_t)f * C + c) = t;
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

// SE excite (second 1x1 conv + sigmoid) and the channel scale in ONE launch: x[f, :, c] *= sigmoid(W2[c, :] . g1[f, :] + b2[c]).
// grid = (C/128, F), block 256.  A workgroup owns 128 channels of one frame: wave w computes the 32 gates w*32 .. w*32+31, 16 at a time (each lane keeps its 16-wide
// k slice of g1 per 1024-wide chunk in registers and streams the weight rows past it, four rows per batch; the 16 lane-partials meet in a
// reduce-scatter butterfly), the gates go through LDS, then the 256 threads sweep the frame's H*W positions, 16 positions x 256 B per pass.  The
// extra W2 traffic (128 rows x rd x 2 B per workgroup, L2 hits) rides in front of a sweep that moves 2 x HW x 256 B; it replaces a launch whose
// 14 - 16 us were latency, not bytes (small_linear_kernel at [F, C] x [C, rd]).
__global__ __launch_bounds__(256, 4) void se_excite_scale_kernel(bf16_t* __restrict__ x, const float* __restrict__ g1, const bf16_t* __restrict__ W2,
                                                              const float* __restrict__ b2, int HW, int C, int rd) {
#pragma clang fp reassociate(off)
    __shared__ float gate_s[128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, f = blockIdx.y;
    const int c0 = blockIdx.x * 128;
    for (int half16 = 0; half16 < 2; ++half16) {                   // 16 gates at a time: 16 accumulators + 8 weight vectors stay under 128 VGPRs
        float tot[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] = 0.f;
        for (int kc = 0; kc < rd; kc += 1024) {
            const int k = kc + lane * 16;
            const bool kin = k < rd;
            f32x4 gv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) gv[t] = kin ? *(const f32x4*)(g1 + (size_t)f * rd + k + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < 16; rb += 4) {
                u32x4 wr[4][2];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + wave * 32 + half16 * 16 + rb + r;
                    const bf16_t* wp = W2 + (size_t)(c < C ? c : C - 1) * rd + (kin ? k : 0);
                    wr[r][0] = *(const u32x4*)wp;
                    wr[r][1] = *(const u32x4*)(wp + 8);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float w[16];
                    unpack8(wr[r][0], w);
                    unpack8(wr[r][1], w + 8);
                    float a = tot[rb + r];
#pragma unroll
                    for (int e = 0; e < 16; ++e) a = __builtin_fmaf(w[e], gv[e >> 2][e & 3], a);
                    tot[rb + r] = kin ? a : tot[rb + r];
                }
            }
        }
        // 16 lane-partials -> lanes 4r .. 4r+3 end up with the wave total of value r: reduce-scatter over lane bits 5..2, full steps over bits 1, 0
#pragma unroll
        for (int m = 32, half = 8; m >= 4; m >>= 1, half >>= 1) {
            const bool up = lane & m;
#pragma unroll
            for (int k = 0; k < half; ++k) {
                const float send = up ? tot[k] : tot[k + half];
                const float keep = up ? tot[k + half] : tot[k];
                tot[k] = keep + __shfl_xor(send, m);
            }
        }
        tot[0] += __shfl_xor(tot[0], 2);
        tot[0] += __shfl_xor(tot[0], 1);
        if (!(lane & 3)) {
            const int r = half16 * 16 + (lane >> 2), c = c0 + wave * 32 + r;
            gate_s[wave * 32 + r] = c < C ? sigmoid_f(tot[0] + (b2 ? b2[c] : 0.f)) : 0.f;
        }
    }
    __syncthreads();
    const int cq = threadIdx.x & 15, pr = threadIdx.x >> 4;      // 16 lanes x 16 B = the 128 channels of one position; 16 positions per pass
    const int c = c0 + cq * 8;
    if (c >= C) return;
    float gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gq[e] = gate_s[cq * 8 + e];
    bf16_t* xf = x + (size_t)f * HW * C + c;
    int p = pr;
    for (; p + 48 < HW; p += 64) {                                 // four passes in flight
        u32x4 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = *(const u32x4*)(xf + (size_t)(p + t * 16) * C);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a[8];
            unpack8(v[t], a);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] *= gq[e];
            *(u32x4*)(xf + (size_t)(p + t * 16) * C) = pack8(a);
        }
    }
    for (; p < HW; p += 16) {
        float a[8];
        unpack8(*(const u32x4*)(xf + (size_t)p * C), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= gq[e];
        *(u32x4*)(xf + (size_t)p * C) = pack8(a);
    }
}

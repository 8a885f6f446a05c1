// One-time weight re-layout kernels (checkpoint layout -> the layouts the compute kernels read), so that a host without
// PyTorch can prepare the weights through the C ABI (include/vl2hip.h vl2_pack_*).  All device -> device, HBM-bound, run once
// per model load.  The Python host (videollama2_amd/weights.py) does the same re-layouts with tensor ops; tests assert that
// both give the same bytes (fp32 sums: to rounding).
//   pack_fold_norm_kernel : W' = bf16(W * g) per input column, s[n] = sum_k W'[n][k], t[n] = sum_k W[n][k] * beta[k] + c[n]
//                           (LayerNorm / RMSNorm affine folded into the following linear layer: k_gemm.h norm-carrying GEMMs)
//   pack_gate_up_kernel   : gate / up [I, D] -> [2I, D] in blocks of 64 rows = 32 gate rows then 32 up rows (VL2_GEMM_SWIGLU)
//   pack_permute_kernel   : [A][B][C] -> [A][C][B] (Conv3d [Co][Ci][2*2*2] -> [Co][tap][Ci]; depthwise [C][9] -> [9][C]),
//                           bf16 or fp32 output
//   pack_pad_rows_kernel  : rows of `cs` elements -> rows of `cd` >= cs elements, zero filled (patch weight K 588 -> 640;
//                           SigLIP head_dim 72 -> 96 and MLP 4304 -> 4352 paddings: a "row" is whatever block is padded)
//   pack_cvt_f32_kernel   : bf16 -> fp32 (norm weights, biases: exact)
#pragma once
#include "dev_common.h"

// one wave per output row n; grid = ceil(N / 4), block 256
__global__ __launch_bounds__(256) void pack_fold_norm_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ g,
                                                             const bf16_t* __restrict__ beta, const bf16_t* __restrict__ c,
                                                             bf16_t* __restrict__ Wp, float* __restrict__ s, float* __restrict__ t,
                                                             int N, int K, int ldw) {
#pragma clang fp reassociate(off)
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bf16_t* w = W + (size_t)n * ldw;
    float ss = 0.f, tt = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wf = bf2f(w[k]);
        const bf16_t wr = f2bf(wf * bf2f(g[k]));
        Wp[(size_t)n * K + k] = wr;
        ss += bf2f(wr);
        if (beta) tt = __builtin_fmaf(wf, bf2f(beta[k]), tt);
    }
    ss = wave_sum(ss);
    tt = wave_sum(tt);
    if (lane == 0) {
        s[n] = ss;
        if (t) t[n] = tt + (c ? bf2f(c[n]) : 0.f);
    }
}

// out row r: block = r / 64, j = r % 64: j < 32 -> gate[block * 32 + j], else up[block * 32 + j - 32]; grid = 2I, block 128
__global__ __launch_bounds__(128) void pack_gate_up_kernel(const bf16_t* __restrict__ gate, const bf16_t* __restrict__ up,
                                                           bf16_t* __restrict__ out, int D) {
    const int r = blockIdx.x, blk = r >> 6, j = r & 63;
    const bf16_t* src = (j < 32 ? gate + (size_t)(blk * 32 + j) * D : up + (size_t)(blk * 32 + j - 32) * D);
    for (int c = threadIdx.x * 8; c < D; c += 1024) *(u32x4*)(out + (size_t)r * D + c) = *(const u32x4*)(src + c);
}

// in [A][B][C] bf16 -> out [A][C][B]; F32: fp32 output.  grid = (ceil(B*C / 256), A)
template <bool F32>
__global__ __launch_bounds__(256) void pack_permute_kernel(const bf16_t* __restrict__ in, void* __restrict__ out, int B, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;               // index into the OUTPUT [C][B] plane
    if (i >= B * C) return;
    const int cc = i / B, bb = i - cc * B;
    const bf16_t v = in[((size_t)blockIdx.y * B + bb) * C + cc];
    const size_t o = (size_t)blockIdx.y * B * C + i;
    if (F32) ((float*)out)[o] = bf2f(v);
    else ((bf16_t*)out)[o] = v;
}

// rows x cs -> rows x cd (cd >= cs), zero filled; grid = rows, block 256
__global__ __launch_bounds__(256) void pack_pad_rows_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, long cs, long cd) {
    const size_t r = blockIdx.x;
    for (long c = threadIdx.x; c < cd; c += 256) out[r * cd + c] = c < cs ? in[r * cs + c] : (bf16_t)0;
}

__global__ __launch_bounds__(256) void pack_cvt_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = bf2f(in[i]);
}

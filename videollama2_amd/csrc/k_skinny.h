// Skinny-M GEMM for batched decode:  C[M <= 64, N] = epilogue( A[M,K] . W[N,K]^T ).
//
// With 5..64 tokens per step the projections are still bound by streaming the weights, but the tiled GEMM kernels put only
// N/64 .. N/128 workgroups on them (a workgroup streams ~55 GB/s through LDS-DMA) and the multi-row GEMV spends its time in
// per-row LDS reads.  Here the weights are streamed the GEMV way -- straight from HBM to registers, every lane 16 contiguous
// bytes of one weight row, eight loads in flight -- and the loaded vector IS the B operand of v_mfma_f32_16x16x32_bf16
// (B[k = (lane>>4)*8 + j][n = lane&15] = W[n][k]): one wave owns 16 output columns and multiplies each weight vector against
// all M <= 64 token rows (x chunk in LDS as the A operand).  K is split over gridDim.y workgroups so that ~4096 waves are in
// flight whatever N is; every workgroup writes its fp32 partial [M, 64] and `skinny_reduce_kernel` sums the slices in order
// (deterministic) and applies bias / residual / SwiGLU / output conversion.
#pragma once
#include "dev_common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
    const bf16_t* A;        // [M, lda]
    const bf16_t* W;        // [N, ldw]
    float* part;            // [KS][Mp][N] fp32 partial sums, Mp = 16 * MT
    int M, N, K, lda, ldw;
    int kslice, kchunk;     // K per workgroup (multiple of kchunk), K per LDS chunk (multiple of 32)
};

// grid = (N/64, KS), block 256; dynamic LDS = Mp * (kchunk + 8) * 2 bytes
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    constexpr int Mp = 16 * MT;
    bf16_t* xs = (bf16_t*)vl2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 64 + wave * 16;
    const int k0 = blockIdx.y * p.kslice;
    const int pitch = p.kchunk + 8;                       // +16 B per row: the 16 rows of a fragment read hit different banks
    const int l15 = lane & 15, lg = lane >> 4;
    const bf16_t* wrow = p.W + (size_t)(n0 + l15) * p.ldw + k0 + lg * 8;

    f32x4v acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int cvec = p.kchunk >> 3;                       // 16-B vectors per x row per chunk
    for (int kc = 0; kc < p.kslice; kc += p.kchunk) {
        const int nstep = p.kchunk >> 5;                  // 32-deep MFMA steps in this chunk
        // the chunk's first weight vectors do not depend on x: request them before staging x
        u32x4 wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nstep) wv[i] = __builtin_nontemporal_load((const u32x4*)(wrow + kc + i * 32));
        __syncthreads();                                  // previous chunk's fragment reads are done
        for (int e = tid; e < Mp * cvec; e += 256) {
            const int r = e / cvec, c = e - r * cvec;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (r < p.M) v = *(const u32x4*)(p.A + (size_t)r * p.lda + k0 + kc + c * 8);
            *(u32x4*)(xs + (size_t)r * pitch + c * 8) = v;
        }
        __syncthreads();
        for (int s0 = 0; s0 < nstep; s0 += 8) {
            if (s0) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (s0 + i < nstep) wv[i] = __builtin_nontemporal_load((const u32x4*)(wrow + kc + (s0 + i) * 32));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (s0 + i < nstep) {
                    const bf16x8 bfrag = __builtin_bit_cast(bf16x8, wv[i]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const bf16x8 afrag = *(const bf16x8*)(xs + (size_t)(m * 16 + l15) * pitch + (s0 + i) * 32 + lg * 8);
                        acc[m] = VL2_MFMA16(afrag, bfrag, acc[m]);
                    }
                }
            }
        }
    }
    // D[row = (lane>>4)*4 + r][col = lane&15]
    float* dst = p.part + (size_t)blockIdx.y * Mp * p.N;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)(m * 16 + lg * 4 + r) * p.N + n0 + l15] = acc[m][r];
}

struct SkinnyReduceArgs {
    const float* part;      // [KS][Mp][N]
    void* C;                // bf16 or fp32 [M, ldc]
    const float* bias;      // [N] or null
    const bf16_t* res;      // [M, ldres] or null
    int M, Mp, N, KS, ldc, ldres;
};

// one thread per 4 consecutive output columns of one row; SWIGLU: W packed in 64-row blocks {32 gate, 32 up}
template <bool SWIGLU, bool OUT_F32>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(SkinnyReduceArgs p) {
    const int ncol = SWIGLU ? p.N / 2 : p.N;             // output columns
    const int per_row = ncol >> 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.M * per_row) return;
    const int m = e / per_row, c = (e - m * per_row) * 4;
    const int n = SWIGLU ? (c >> 5) * 64 + (c & 31) : c;  // GEMM column of the (gate) value
    f32x4v g = f32x4v{0.f, 0.f, 0.f, 0.f}, u = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < p.KS; ++ks) {
        const float* src = p.part + ((size_t)ks * p.Mp + m) * p.N + n;
        const f32x4v a = *(const f32x4v*)src;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] += a[j];
        if (SWIGLU) {
            const f32x4v b = *(const f32x4v*)(src + 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] += b[j];
        }
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = SWIGLU ? silu_f(g[j]) * u[j] : g[j];
        if (!SWIGLU && p.bias) o[j] += p.bias[c + j];
        if (p.res) o[j] += bf2f(p.res[(size_t)m * p.ldres + c + j]);
    }
    if (OUT_F32) {
        *(f32x4v*)((float*)p.C + (size_t)m * p.ldc + c) = f32x4v{o[0], o[1], o[2], o[3]};
    } else {
        u32x2 w;
        w[0] = pack2bf(o[0], o[1]);
        w[1] = pack2bf(o[2], o[3]);
        *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + c) = w;
    }
}

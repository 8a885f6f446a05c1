// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )   (W in torch nn.Linear layout)
//
// Replaces the cuBLAS / cuDNN-1x1-conv calls the reference delegates to: CLIP q/k/v/out/fc1/fc2
// (HF:models/clip/modeling_clip.py CLIPAttention/CLIPMLP), the STC 1x1 convs, Conv3d-as-gathered-GEMM and readout
// (videollama2/model/projector.py:153-187), Mistral q/k/v/o/gate/up/down/lm_head (HF:models/mistral/modeling_mistral.py).
//
// Shape: 128x128 block tile, BK=64, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.
// Staging: LDS-DMA (global_load_lds, 16 B/lane) into a double-buffered 2 x (16 KiB A + 16 KiB W) image.
// The image is lane-linear (LDS-DMA writes base + lane*16), so the bank swizzle lives on the per-lane SOURCE
// address and on the ds_read_b128 address (guide rule 21): a 256-B LDS bank row holds two 128-B tile rows =
// 16 slots of 16 B; slot s of bank row R is stored at s ^ (R & 15)  -> conflict-free fragment reads.
// Epilogue: accumulators -> wave-private fp32 LDS patch -> row-contiguous 16-B stores with fused
// bias / activation / residual / SwiGLU / row remap.
#pragma once
#include "dev_common.h"

struct GemmArgs {
    const bf16_t* A;      // [M, lda] (or row pool for the gathered form)
    const bf16_t* W;      // [N, ldw]
    void* C;              // bf16 or fp32 [*, ldc]
    const float* bias;    // [N] or null
    const bf16_t* res;    // residual rows or null
    const int* a_idx;     // GATHER: [K/seg_k][M] source row per (segment, m); <0 = zero row
    const bf16_t* zero_row;  // GATHER: >= seg_k zeros
    int M, N, K;
    int lda, ldw, ldc, ldres;
    int seg_k;
    int out_grp, out_grp_pad, out_row_off;   // out_row = m + (m / out_grp) * out_grp_pad + out_row_off   (out_grp > 0)
    int res_row_mod, res_row_off;            // res_row = res_row_mod > 0 ? m % res_row_mod + res_row_off : out_row
    int tiles_m, tiles_n;
};

enum { ACT_NONE = 0, ACT_QGELU = 1, ACT_GELU = 2, ACT_SILU = 3 };

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_LDS_BYTES 65536

__device__ __forceinline__ int gemm_lds_off(int row, int chunk) {   // byte offset of 16-B chunk `chunk` of tile row `row`
    const int R = row >> 1;
    const int s = ((row & 1) << 3) | chunk;
    return ((R << 4) + (s ^ (R & 15))) << 4;
}

// One wave-private fp32 patch ep[32][68] (64 GEMM columns) -> global rows: fused bias / activation / SwiGLU / residual /
// row remap, 16-B stores.  m_base = first row of the patch, n_base = first (un-halved) GEMM column of the patch.
template <int ACT, bool SWIGLU, bool OUT_F32>
__device__ __forceinline__ void gemm_store_patch(const GemmArgs& p, const float* ep, int m_base, int n_base, int lane) {
        constexpr int LPR = SWIGLU ? 4 : 8;            // lanes per row
        constexpr int RPP = 64 / LPR;                  // rows per pass
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; ++pass) {
            const int row = pass * RPP + lane / LPR;
            const int cg = (lane % LPR) * 8;
            const int m = m_base + row;
            if (m < p.M) {
                float v[8];
                const int nfull = n_base + cg;             // column in the (un-halved) GEMM N space
                if (SWIGLU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float g = ep[row * 68 + cg + j], u = ep[row * 68 + 32 + cg + j];
                        v[j] = silu_f(g) * u;
                    }
                } else {
                    const f32x4 x0 = *(const f32x4*)(ep + row * 68 + cg), x1 = *(const f32x4*)(ep + row * 68 + cg + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = x0[j]; v[4 + j] = x1[j]; }
                    if (p.bias) {
                        const f32x4 b0 = *(const f32x4*)(p.bias + nfull), b1 = *(const f32x4*)(p.bias + nfull + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ACT == ACT_QGELU) v[j] = quick_gelu_f(v[j]);
                        if (ACT == ACT_GELU) v[j] = gelu_erf_f(v[j]);
                        if (ACT == ACT_SILU) v[j] = silu_f(v[j]);
                    }
                }
                const int n = SWIGLU ? (n_base >> 1) + cg : nfull;
                const int orow = p.out_grp > 0 ? m + (m / p.out_grp) * p.out_grp_pad + p.out_row_off : m;
                if (p.res) {
                    const int rrow = p.res_row_mod > 0 ? (m % p.res_row_mod) + p.res_row_off : orow;
                    const u32x4 rv = *(const u32x4*)(p.res + (size_t)rrow * p.ldres + n);
                    float rf[8];
                    unpack8(rv, rf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += rf[j];
                }
                if (OUT_F32) {
                    float* c = (float*)p.C + (size_t)orow * p.ldc + n;
                    f32x4 o0, o1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { o0[j] = v[j]; o1[j] = v[4 + j]; }
                    *(f32x4*)c = o0;
                    *(f32x4*)(c + 4) = o1;
                } else {
                    *(u32x4*)((bf16_t*)p.C + (size_t)orow * p.ldc + n) = pack8(v);
                }
            }
        }
}

template <int ACT, bool SWIGLU, bool OUT_F32, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: each XCD walks a contiguous run of tiles, grouped 8 tile-rows deep
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int grp_sz = 8 * p.tiles_n;
    const int first_m = (t / grp_sz) * 8;
    const int gm = (p.tiles_m - first_m) < 8 ? (p.tiles_m - first_m) : 8;
    const int tm = first_m + (t % grp_sz) % gm, tn = (t % grp_sz) / gm;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    // per-thread staging coordinates (4 x 16 B for A, 4 x 16 B for W per K-tile)
    int st_row[4], st_chk[4];
    const bf16_t* a_src[4];
    const bf16_t* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = ((i * 4 + wave) << 6) + lane;
        const int R = slot >> 4, sp = slot & 15, s = sp ^ (R & 15);
        st_row[i] = 2 * R + (s >> 3);
        st_chk[i] = s & 7;
        int am = m0 + st_row[i];
        am = am < p.M ? am : p.M - 1;
        a_src[i] = GATHER ? nullptr : p.A + (size_t)am * p.lda + st_chk[i] * 8;
        st_row[i] = am;
        b_src[i] = p.W + (size_t)(n0 + 2 * R + (s >> 3)) * p.ldw + st_chk[i] * 8;
    }

    auto stage = [&](int buf, int kt) {
        unsigned char* As = vl2_smem + buf * 32768;
        unsigned char* Bs = As + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src;
            if (GATHER) {
                const int k = kt * GEMM_BK;
                const int seg = k / p.seg_k, koff = k - seg * p.seg_k;
                const int r = p.a_idx[(size_t)seg * p.M + st_row[i]];
                src = (r < 0 ? p.zero_row : p.A + (size_t)r * p.lda) + koff + st_chk[i] * 8;
            } else {
                src = a_src[i] + kt * GEMM_BK;
            }
            glds16(src, As + ((i * 4 + wave) << 10));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(b_src[i] + kt * GEMM_BK, Bs + ((i * 4 + wave) << 10));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = p.K / GEMM_BK;
    const int frow = lane & 31, fchk = lane >> 5;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) stage((kt + 1) & 1, kt + 1);
        const unsigned char* As = vl2_smem + (kt & 1) * 32768;
        const unsigned char* Bs = As + 16384;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const bf16x8*)(As + gemm_lds_off(wm * 64 + i * 32 + frow, ks * 2 + fchk));
                bfr[i] = *(const bf16x8*)(Bs + gemm_lds_off(wn * 64 + i * 32 + frow, ks * 2 + fchk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: acc (col = lane&31, rows (r&3)+8(r>>2)+4(lane>>5)) -> fp32 LDS patch [32][68] per wave -> rows
    float* ep = (float*)vl2_smem + wave * (32 * 68);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ep[row * 68 + ni * 32 + (lane & 31)] = acc[mi][ni][r];
            }
        __syncthreads();
        gemm_store_patch<ACT, SWIGLU, OUT_F32>(p, ep, m0 + wm * 64 + mi * 32, n0 + wn * 64, lane);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gemm2: 128 x 256 block tile, BK = 32, 4 waves (2x2), wave tile 64 x 128 = 2x4 v_mfma_f32_32x32x16_bf16 (128 acc VGPRs),
// double-buffered 2 x (8 KiB A + 16 KiB W) = 48 KiB LDS -> THREE workgroups per CU (<= 168 VGPRs).
// Why: at 128x128 the L2 -> LDS operand stream is 15.3 B/kFLOP, i.e. ~17 TB/s chip-wide at 1.1 PF/s -- half of the
// ~34 TB/s aggregate L2 bandwidth -- and a wave reads 1 KiB of LDS per MFMA.  128x256 cuts the operand stream to 3/4 and
// the LDS reads to 0.75 KiB per MFMA at the same 16-MFMAs-per-barrier cadence; 3 resident workgroups hide the barrier.
// LDS image: 64-B tile rows, 4 per 256-B bank row; chunk c of row r is stored at c ^ ((r>>2)&3) (conflict-free b128 reads).
#define GEMM2_BM 128
#define GEMM2_BN 256
#define GEMM2_BK 32
#define GEMM2_LDS_BYTES 49152

__device__ __forceinline__ int gemm2_lds_off(int row, int chunk) {
    return (((row >> 2) << 4) + ((row & 3) << 2) + (chunk ^ ((row >> 2) & 3))) << 4;
}

template <int ACT, bool SWIGLU, bool OUT_F32, bool GATHER>
__global__ __launch_bounds__(256, 3) void gemm2_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int grp_sz = 8 * p.tiles_n;
    const int first_m = (t / grp_sz) * 8;
    const int gm = (p.tiles_m - first_m) < 8 ? (p.tiles_m - first_m) : 8;
    const int tm = first_m + (t % grp_sz) % gm, tn = (t % grp_sz) / gm;
    const int m0 = tm * GEMM2_BM, n0 = tn * GEMM2_BN;

    // staging: A tile 128 rows x 4 chunks = 2 LDS-DMA per thread; W tile 256 rows x 4 chunks = 4 per thread.
    // 32-bit element offsets from the (uniform) base pointers keep the address state in few VGPRs.
    const int slot0 = (wave << 6) + lane;                          // i-th LDS-DMA covers slot0 + 256*i  (64 rows per i)
    const int srow = 4 * (slot0 >> 4) + ((slot0 & 15) >> 2);
    const int schk = (slot0 & 3) ^ ((slot0 >> 4) & 3);             // (R & 3) is the same for every i (R advances by 16)
    int a_row[2];
    unsigned a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int am = m0 + srow + 64 * i;
        am = am < p.M ? am : p.M - 1;
        a_row[i] = am;
        a_off[i] = (unsigned)am * (unsigned)p.lda + schk * 8;
    }
    const unsigned b_off = (unsigned)(n0 + srow) * (unsigned)p.ldw + schk * 8;
    const unsigned b_step = 64u * (unsigned)p.ldw;

    auto stage = [&](int buf, int kt) {
        unsigned char* As = vl2_smem + buf * 24576;
        unsigned char* Bs = As + 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src;
            if (GATHER) {
                const int k = kt * GEMM2_BK;
                const int seg = k / p.seg_k, koff = k - seg * p.seg_k;
                const int r = p.a_idx[(size_t)seg * p.M + a_row[i]];
                src = (r < 0 ? p.zero_row : p.A + (size_t)r * p.lda) + koff + schk * 8;
            } else {
                src = p.A + a_off[i] + kt * GEMM2_BK;
            }
            glds16(src, As + ((i * 4 + wave) << 10));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(p.W + b_off + i * b_step + kt * GEMM2_BK, Bs + ((i * 4 + wave) << 10));
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = p.K / GEMM2_BK;
    const int frow = lane & 31, fchk = lane >> 5;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) stage((kt + 1) & 1, kt + 1);
        const unsigned char* As = vl2_smem + (kt & 1) * 24576;
        const unsigned char* Bs = As + 8192;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(As + gemm2_lds_off(wm * 64 + i * 32 + frow, ks * 2 + fchk));
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {                      // two W fragments live at a time (register budget 168)
                bf16x8 bfr[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bfr[j] = *(const bf16x8*)(Bs + gemm2_lds_off(wn * 128 + (jh * 2 + j) * 32 + frow, ks * 2 + fchk));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][jh * 2 + j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: four 32 x 64 patches per wave (2 row blocks x 2 column halves), same store path as gemm_bf16_kernel
    float* ep = (float*)vl2_smem + wave * (32 * 68);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    ep[row * 68 + ni * 32 + (lane & 31)] = acc[mi][nh * 2 + ni][r];
                }
            __syncthreads();
            gemm_store_patch<ACT, SWIGLU, OUT_F32>(p, ep, m0 + wm * 64 + mi * 32, n0 + wn * 128 + nh * 64, lane);
            __syncthreads();
        }
}

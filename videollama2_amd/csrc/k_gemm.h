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

// a use the optimiser cannot move: pins the (pure) producers of a, b before this point.  The CPU test build defines it empty.
#ifndef VL2_PIN2
#define VL2_PIN2(a, b) asm volatile("" :: "v"(a), "v"(b))
#endif

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
    // stream-K form (gemm_sk_bf16_kernel): fp32 partial-tile workspace [grid][64][256], one flag per workgroup (zeroed by
    // the launcher before every launch), K-tile iterations per workgroup
    float* sk_ws;
    int* sk_flags;
    int sk_per;
    int idx_ld;              // GATHER: row stride of a_idx (= M of the whole call; a row chunk keeps the full table)
    // ---- normalisation carried through the GEMM chain (see "norm-carrying GEMMs" below)
    float* stats_out;        // [rows][stats_out_np][2] fp32 (sum, sum of squares) of the bf16-rounded OUTPUT rows per 64-column block, or null
    int stats_out_np;        // partial blocks per row of stats_out (= N / 64 of the whole call)
    const float* stats_in;   // [rows][stats_in_np][2] statistics of the A rows written by the GEMM that produced A, or null
    int stats_in_np;         // = K / 64
    int norm;                // 0 none, 1 RMSNorm, 2 LayerNorm of the A rows, applied algebraically in the epilogue
    float norm_eps;
    const float* w_colsum;   // LayerNorm: s[n] = sum_k W[n][k] (fp32, of the bf16 weights as packed) [N]
    const float* row_norm;   // [rows][2] = (mean, rstd) of the A rows, already reduced (row_norm_finalize_kernel); overrides stats_in
    const float* col_scale;  // fp8 form: [N] one multiplier per output column (the weight rows' scales), applied behind the row scale; else null
    unsigned* tile_ctr;      // persistent form (k_gemm6.h): two zeroed words {tiles handed out, workgroups finished}, re-armed by the kernel; null = static walk
    int tile_first_dyn;      // persistent form: 1 = the FIRST tile of a workgroup is drawn from the counter too (a workgroup that starts late finds no work)
    // ---- producer-side finalize (gemm_rows_ticket below): with stats_out, the workgroup that stores the LAST column tile of a row block reduces the
    // block's partials to (mean, rstd) itself -- the row_norm_finalize launch between a producer and its consumer disappears
    float* row_norm_out;     // [rows][2] (mean, rstd) of the OUTPUT rows, or null
    unsigned* row_ticket;    // one zeroed word per row tile of THIS launch (index = the kernel's tile row), re-armed by the last arriver
    int norm_out;            // 1 RMSNorm (mean = 0), 2 LayerNorm
    float norm_out_eps;
    int tile_group;          // k_gemm9.h: depth of the row-tile group that shares a W panel on an XCD; 0 = the kernel's own (4)
};

// ---- norm-carrying GEMMs ------------------------------------------------------------------------------------------------
// y = Norm(x) W^T with the norm's affine part folded into the weights at pack time (W' = W * g per input column, t = W b + c):
//   RMSNorm:    y[m][n] = rstd_m * (x W'^T)[m][n]                                  (HF:modeling_mistral.py MistralRMSNorm)
//   LayerNorm:  y[m][n] = rstd_m * ((x W'^T)[m][n] - mean_m * s[n]) + t[n],   s[n] = sum_k W'[n][k]
//                                                                                  (HF:modeling_clip.py layer_norm1/2 -> q/k/v, fc1)
// i.e. the MFMA main loop runs on the RAW residual-stream rows and the normalisation is two FMAs per output element in the
// epilogue: the standalone norm kernels (a full HBM read + write of the activation between every two GEMMs: 1.2 ms of the
// 39.6 ms T=16 forward, profiles/r01_kernel_stats_bench_default_late.csv) disappear.  The row statistics come from the GEMM
// that WROTE x (o-proj / down-proj / out_proj / fc2, all with the residual add fused): its epilogue emits, per row and per
// 64-column patch, (sum, sum of squares) of the values exactly as stored (after the bf16 rounding) -- `stats_out`; the
// consumer's workgroup reduces the K/64 partials of its BM rows once, in its prologue while the first LDS-DMA slabs are in
// flight (`gemm_row_stats`), and parks (mean, rstd) in LDS for the epilogue.  Fixed partial layout + fixed reduction order:
// a row's statistics are the same bits whichever kernel / grid produced them.
// (sum, sum of squares) of a row from its `np` per-64-column partials, in the association of row_norm_finalize_kernel (k_norm.h:
// eight strided sums -- accumulator i takes partials i, i + 8, ... in increasing order -- folded like octet_sum), so a GEMM that
// reduces the partials itself normalises with the same (mean, rstd) bits as one that is handed the finalize kernel's table.
// The loads go out in batches of eight 16-B vectors (one memory round trip per 16 partials), not one per loop trip.
__device__ __forceinline__ f32x2 row_partials_reduce(const float* sp, int np) {
#pragma clang fp reassociate(off)
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
    const int nv = np >> 1;                                       // 16-B vectors (two partials each); np is even (K % 128 == 0)
    for (int c0 = 0; c0 < nv; c0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (c0 + k < nv) ? *(const f32x4*)(sp + 4 * (c0 + k)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (c0 + k < nv) {                                    // partials 2 (c0 + k), + 1  ->  accumulators (2 k) & 7, + 1
                s[(2 * k) & 7] += v[k][0];
                q[(2 * k) & 7] += v[k][1];
                s[(2 * k + 1) & 7] += v[k][2];
                q[(2 * k + 1) & 7] += v[k][3];
            }
    }
    if (np & 1) { s[(np - 1) & 7] += sp[2 * (np - 1)]; q[(np - 1) & 7] += sp[2 * (np - 1) + 1]; }
    f32x2 r;
    r[0] = ((s[0] + s[7]) + (s[1] + s[6])) + ((s[2] + s[5]) + (s[3] + s[4]));
    r[1] = ((q[0] + q[7]) + (q[1] + q[6])) + ((q[2] + q[5]) + (q[3] + q[4]));
    return r;
}
__device__ __forceinline__ f32x2 gemm_row_stats(const GemmArgs& p, int m0, int t, int BM) {
#pragma clang fp reassociate(off)
    f32x2 r = {0.f, 1.f};
    if (p.norm && t < BM) {
        int m = m0 + t;
        m = m < p.M ? m : p.M - 1;
        if (p.row_norm) return *(const f32x2*)(p.row_norm + 2 * (size_t)m);      // reduced once per GEMM call (row_norm_finalize_kernel)
        const f32x2 sq = row_partials_reduce(p.stats_in + (size_t)m * p.stats_in_np * 2, p.stats_in_np);
        const float inv = 1.0f / (float)p.K;                                     // same expressions as row_norm_finalize_kernel
        if (p.norm == 2) {
            const float mean = sq[0] * inv;
            r[0] = mean;
            r[1] = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, sq[1] * inv), 0.f) + p.norm_eps);
        } else {
            r[1] = rsqrtf(sq[1] * inv + p.norm_eps);
        }
    }
    return r;
}
// (mean, rstd) of the tile's rows -> LDS table `tab` [BM][2]; call before the barrier that precedes the first gemm_store_patch
__device__ __forceinline__ void gemm_park_row_stats(const GemmArgs& p, float* tab, const f32x2& r, int t, int BM) {
    if (p.norm && t < BM) { tab[2 * t] = r[0]; tab[2 * t + 1] = r[1]; }
}

// One (sum, sum of squares) partial of `stats_out`.  With the producer-side finalize the partial goes out WRITE-THROUGH (an agent-scope relaxed
// atomic store = `global_store_dwordx2 ... sc1`): the workgroup that reduces it may sit on another XCD, whose L2 is not coherent with this one's,
// and a release fence (buffer_wbl2) in every tile was measured to lose against the launch it replaces (k_decode.h attn_decode_kernel<FUSED>).
__device__ __forceinline__ void gemm_stat_put(const GemmArgs& p, float* dst, float s, float q) {
    if (p.row_norm_out) {
        const f32x2 v = {s, q};
        __hip_atomic_store((uint64_t*)dst, __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        dst[0] = s;
        dst[1] = q;
    }
}
// ---- producer-side finalize.  Called by every thread of the workgroup at the very end of a tile whose epilogue emitted `stats_out` partials
// (tile row `tm`, rows m0 .. m0 + BM - 1, NT threads).  Every wave drains its write-through partial stores (s_waitcnt vmcnt(0): acknowledged =
// visible at the memory side), the workgroup takes a ticket on its row tile's counter, and the one that draws the LAST ticket of the row block
// (tiles_n of them) reduces the block's N / 64 partials per row to (mean, rstd) -- eight lanes per row, lane i sums partials i, i + 8, ... and
// octet_sum folds them: row_norm_finalize_kernel's (k_norm.h) association and expressions, i.e. the SAME bits as the separate launch -- reading
// them past its own L2 (agent-scope loads).  The counter is re-armed by that workgroup (every other arrival has happened), so one zeroed block
// of counters serves every GEMM of a stream.  No spin anywhere: nobody waits for anybody.
template <int NT>
__device__ __forceinline__ void gemm_rows_ticket(const GemmArgs& p, int tm, int m0, int BM, int tid) {
#pragma clang fp reassociate(off)
    if (!p.row_norm_out || !p.stats_out) return;
    __shared__ int s_rows_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.row_ticket + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)p.tiles_n - 1u;
        if (last) __hip_atomic_store(p.row_ticket + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_rows_last = last;
    }
    __syncthreads();
    if (!s_rows_last) return;
    const int np = p.stats_out_np, i8 = tid & 7;
    const float inv = 1.0f / (float)(np * 64);
    for (int r0 = 0; r0 < BM; r0 += NT / 8) {
        const int r = r0 + (tid >> 3), m = m0 + r;
        const int mc = m < p.M ? m : p.M - 1;                       // whole octets stay active for the DPP reduction
        const float* sp = p.stats_out + (size_t)mc * np * 2;
        float sum = 0.f, sq = 0.f;
        for (int i = i8; i < np; i += 8) {
            const f32x2 v = __builtin_bit_cast(f32x2, __hip_atomic_load((const uint64_t*)(sp + 2 * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            sum += v[0];
            sq += v[1];
        }
        sum = octet_sum(sum);
        sq = octet_sum(sq);
        if (r < BM && m < p.M && i8 == 0) {
            float mean = 0.f, rstd;
            if (p.norm_out == 2) {
                mean = sum * inv;
                rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, sq * inv), 0.f) + p.norm_out_eps);
            } else {
                rstd = rsqrtf(sq * inv + p.norm_out_eps);
            }
            p.row_norm_out[2 * (size_t)m] = mean;
            p.row_norm_out[2 * (size_t)m + 1] = rstd;
        }
    }
}

enum { ACT_NONE = 0, ACT_QGELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_GELU_TANH = 5 };

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
// REMAP = the row-remap / residual-row-modulo form (runtime integer divisions) -- only the patch-embed GEMM needs it
template <int ACT, bool SWIGLU, bool OUT_F32, bool REMAP = false>
__device__ __forceinline__ void gemm_store_patch(const GemmArgs& p0, const float* ep, int m_base, int n_base, int lane,
                                                 const float* rowtab = nullptr, int mt_base = 0) {
        // every GEMM kernel inlines this epilogue, and a row must come out with the same bits whichever kernel its shape selects
        // (sharded == unsharded, batched == one by one): (acc + bias) + residual stays in that order in every instantiation.
        // rowtab: LDS table of (mean, rstd) of the tile's rows (p.norm != 0), indexed by the row's offset in the tile.
#pragma clang fp reassociate(off)
        const GemmArgs& p = p0;
        constexpr int LPR = SWIGLU ? 4 : 8;            // lanes per row
        constexpr int RPP = 64 / LPR;                  // rows per pass
        constexpr int NP = 32 / RPP;                   // row passes per patch (4, SwiGLU 2)
        // a lane keeps its 8 columns through all passes: the per-column vectors (bias, LayerNorm column sums) are loaded once
        // per patch, not once per row pass (the stores to C in the loop keep the compiler from hoisting them itself)
        const int cg = (lane % LPR) * 8;
        const int nfull = n_base + cg;                 // column in the (un-halved) GEMM N space
        const int n = SWIGLU ? (n_base >> 1) + cg : nfull;
        f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0, cs0 = bias0, cs1 = bias0;
        if (!SWIGLU && p.bias) { bias0 = *(const f32x4*)(p.bias + nfull); bias1 = *(const f32x4*)(p.bias + nfull + 4); }
        if (!SWIGLU && p.norm == 2) { cs0 = *(const f32x4*)(p.w_colsum + nfull); cs1 = *(const f32x4*)(p.w_colsum + nfull + 4); }
        // fp8 form only (null otherwise: the 16-bit paths keep their bits): the weight rows' scales of this lane's 8 columns (SwiGLU: gate | up)
        f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sc2 = sc0, sc3 = sc0;
        if (p.col_scale) {
            sc0 = *(const f32x4*)(p.col_scale + nfull); sc1 = *(const f32x4*)(p.col_scale + nfull + 4);
            if (SWIGLU) { sc2 = *(const f32x4*)(p.col_scale + nfull + 32); sc3 = *(const f32x4*)(p.col_scale + nfull + 36); }
        }
        // Round 3 (profiles/r03_experiments.md): the epilogue was 7-12 us of every tile (22-36 % of a K = 1024 GEMM), a chain of
        // LDS write -> read -> RESIDUAL LOAD (an L2 / HBM round trip under `if (live)`, so never hoisted) -> store per row pass.
        // The residual rows of all passes are now requested up front, unconditionally (rows past M clamp to M - 1 and are dropped
        // by the store predicate): one memory round trip per patch instead of one per pass.  Same arithmetic, same bits.
        int orow_[NP];
        u32x4 rv_[NP];
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int m = m_base + pass * RPP + lane / LPR;
            const int mc = m < p.M ? m : p.M - 1;
            orow_[pass] = (REMAP && p.out_grp > 0) ? mc + (mc / p.out_grp) * p.out_grp_pad + p.out_row_off : mc;
            if (p.res) {
                const int rrow = (REMAP && p.res_row_mod > 0) ? (mc % p.res_row_mod) + p.res_row_off : orow_[pass];
                rv_[pass] = *(const u32x4*)(p.res + (size_t)rrow * p.ldres + n);
            }
        }
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int row = pass * RPP + lane / LPR;
            const int m = m_base + row;
            const bool live = m < p.M;
            float st_s = 0.f, st_q = 0.f;
            const int orow = orow_[pass];
            if (live) {
                float v[8];
                float mu = 0.f, rs = 1.f;
                if (p.norm) { mu = rowtab[2 * (mt_base + row)]; rs = rowtab[2 * (mt_base + row) + 1]; }
                if (SWIGLU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float g = ep[row * 68 + cg + j], u = ep[row * 68 + 32 + cg + j];
                        if (p.norm) { g *= rs; u *= rs; }                     // RMSNorm only (no mean / shift term)
                        if (p.col_scale) { g *= (j < 4 ? sc0[j & 3] : sc1[j & 3]); u *= (j < 4 ? sc2[j & 3] : sc3[j & 3]); }
                        v[j] = silu_f(g) * u;
                    }
                } else {
                    const f32x4 x0 = *(const f32x4*)(ep + row * 68 + cg), x1 = *(const f32x4*)(ep + row * 68 + cg + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = x0[j]; v[4 + j] = x1[j]; }
                    if (p.norm == 2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] = __builtin_fmaf(-mu, cs0[j], v[j]) * rs; v[4 + j] = __builtin_fmaf(-mu, cs1[j], v[4 + j]) * rs; }
                    } else if (p.norm == 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= rs;
                    }
                    if (p.col_scale) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] *= sc0[j]; v[4 + j] *= sc1[j]; }
                    }
                    if (p.bias) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] += bias0[j]; v[4 + j] += bias1[j]; }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ACT == ACT_QGELU) v[j] = quick_gelu_f(v[j]);
                        if (ACT == ACT_GELU) v[j] = gelu_erf_f(v[j]);
                        if (ACT == ACT_SILU) v[j] = silu_f(v[j]);
                        if (ACT == ACT_GELU_TANH) v[j] = gelu_tanh_f(v[j]);
                    }
                }
                if (p.res) {
                    float rf[8];
                    unpack8(rv_[pass], rf);
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
                    const u32x4 packed = pack8(v);
                    *(u32x4*)((bf16_t*)p.C + (size_t)orow * p.ldc + n) = packed;
                    if (!SWIGLU && !REMAP && p.stats_out) {       // statistics of the row AS STORED (bf16-rounded)
                        float rf[8];
                        unpack8(packed, rf);
#pragma unroll
                        for (int j = 0; j < 8; ++j) { st_s += rf[j]; st_q = __builtin_fmaf(rf[j], rf[j], st_q); }
                    }
                }
            }
            if (!SWIGLU && !REMAP && !OUT_F32 && p.stats_out) {   // wave-uniform: the 8 lanes of a row fold their partials
                st_s = octet_sum(st_s);
                st_q = octet_sum(st_q);
                if (live && (lane % LPR) == 0) {
                    gemm_stat_put(p, p.stats_out + ((size_t)orow * p.stats_out_np + (n_base >> 6)) * 2, st_s, st_q);
                }
            }
        }
}

// ---- register-resident epilogue for kernels whose MFMAs were issued with the operands SWAPPED (acc = mfma(w_frag, a_frag, acc)):
// the accumulator block then holds C^T -- lane (l & 31) owns ROW m of the 32 x 32 block and its 16 registers are columns
// (r & 3) + 8 (r >> 2) + 4 (l >> 5) -- so the row-major image needs no trip through LDS.
// Round 3 measurements (profiles/r03_experiments.md): the epilogue of a 256 x 256 tile costs 7-12 us (22-36 % of a K = 1024 GEMM).
// It is NOT the stores (ablated: -1.6 us), not HBM (all tiles storing into one L2-resident window: -0.7 us), not the LDS
// bandwidth as such (a first register-resident form with fp32 swaps measured the same): with two waves per SIMD and 128
// outputs per lane it is instruction ISSUE -- every per-element instruction of the epilogue costs ~0.5 us per tile.  So this form
// minimises instructions per element: all arithmetic runs on the accumulator registers where they are (bias / LayerNorm column
// sums are fetched in this lane's column pattern, the residual as 8-byte pieces), the results are packed to bf16 FIRST and one
// v_permlane32_swap per packed register pair (guide T21) then turns the two half-waves' 4-column groups into 8 contiguous columns
// = one 16-B store.  Per 32 x 32 block: 8 cvt_pk + 4 swaps + 2 stores instead of 16 ds_write + 4 ds_read + 8 cvt_pk + 2 stores,
// and no row table, no barrier.  EF >= 0 compiles the runtime flags in (bit 0 bias, 1 RMSNorm, 2 LayerNorm, 3 residual,
// 4 row statistics); EF < 0 tests them at run time.  Arithmetic per element = gemm_store_patch's, same order:
// ((acc [norm]) + bias) -> activation -> + residual -> bf16; statistics from the stored bf16 values in the same association --
// a row's bits do not depend on which epilogue stored it (hash-checked on hardware, scripts/ubench/gemm_lab.hip).
// acc[mi][nj]: 32-row blocks mi (rows m_w0 + 32 mi + (lane & 31)), 32-column blocks nj (columns n_w0 + 32 nj + ...), NJ even.
// four v_permlane32_swap on pk[0..7]: (pk[0], pk[2]), (pk[1], pk[3]), (pk[4], pk[6]), (pk[5], pk[7]) as (vdst, src) -- lanes 32-63 of vdst
// trade places with lanes 0-31 of src; s_nop 1 in front = the wait states a VALU write needs before the swap reads it, s_nop 1 behind
// = the same for the swap's results (the hazard recogniser does not see inside the asm).  (The CPU test build defines the macro itself.)
#ifndef VL2_PERMLANE32_SWAP_8
#define VL2_PERMLANE32_SWAP_8(pk)                                                                                              \
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"                                 \
                 "v_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7\n\ts_nop 1"                                        \
                 : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]), "+v"(pk[4]), "+v"(pk[5]), "+v"(pk[6]), "+v"(pk[7]))
#endif
enum { EF_BIAS = 1, EF_RMS = 2, EF_LN = 4, EF_RES = 8, EF_STATS = 16 };
__device__ __forceinline__ int gemm_ef_code(const GemmArgs& p) {
    return (p.bias ? EF_BIAS : 0) | (p.norm == 1 ? EF_RMS : 0) | (p.norm == 2 ? EF_LN : 0) | (p.res ? EF_RES : 0) | (p.stats_out ? EF_STATS : 0);
}
// rowst[mi] = (mean, rstd) of row m_w0 + 32 mi + (lane & 31) (gemm_tr_row_stats, computed in the kernel's PROLOGUE so that the loads
// of the statistics hide behind the first LDS-DMA slabs)
template <int MI>
__device__ __forceinline__ void gemm_tr_row_stats(const GemmArgs& p, int m_w0, int lane, f32x2 (&rowst)[MI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) rowst[mi] = gemm_row_stats(p, m_w0 + mi * 32, lane & 31, 32);
}
template <int ACT, bool SWIGLU, int MI, int NJ, int EF = -1>
__device__ __forceinline__ void gemm_store_tr(const GemmArgs& p, f32x16 (&acc)[MI][NJ], int m_w0, int n_w0, int lane, const f32x2 (&rowst)[MI]) {
#pragma clang fp reassociate(off)
    const bool has_bias = EF < 0 ? (!SWIGLU && p.bias != nullptr) : (!SWIGLU && (EF & EF_BIAS) != 0);
    const bool norm_rms = EF < 0 ? p.norm == 1 : (EF & EF_RMS) != 0;
    const bool norm_ln = EF < 0 ? (!SWIGLU && p.norm == 2) : (!SWIGLU && (EF & EF_LN) != 0);
    const bool has_res = EF < 0 ? p.res != nullptr : (EF & EF_RES) != 0;
    const bool has_stats = EF < 0 ? (!SWIGLU && p.stats_out != nullptr) : (!SWIGLU && (EF & EF_STATS) != 0);
    const int hi = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m_w0 + mi * 32 + l31;
        const bool live = m < p.M;
        const int mc = live ? m : p.M - 1;
        const float mu = rowst[mi][0], rs = rowst[mi][1];                  // (mean, rstd) of this lane's row (norm-carrying GEMMs)
        bf16_t* crow = (bf16_t*)p.C + (size_t)m * p.ldc;
        const bf16_t* rrow = has_res ? p.res + (size_t)mc * p.ldres : nullptr;
#pragma unroll
        for (int cb = 0; cb < NJ / 2; ++cb) {                                // 64 GEMM columns = one statistics block / SwiGLU block
            float s8[4], q8[4];                                              // statistics of this lane's four 8-column groups
#pragma unroll
            for (int half = 0; half < (SWIGLU ? 1 : 2); ++half) {
                const int nj = cb * 2 + half;
                // un-halved GEMM column / output column of this lane's group g = 0: + 8 g + 4 hi + c
                const int ncol = n_w0 + nj * 32 + 4 * hi;
                const int ocol = SWIGLU ? (n_w0 >> 1) + cb * 32 + 4 * hi : ncol;
                float x[16];
                if (SWIGLU) {                                                // columns [64 cb, +32) gate, [64 cb + 32, +32) up
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float g = acc[mi][nj][r], u = acc[mi][nj + 1][r];
                        if (norm_rms) { g *= rs; u *= rs; }
                        x[r] = silu_f(g) * u;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = acc[mi][nj][r];
                    if (norm_ln) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 cs = *(const f32x4*)(p.w_colsum + ncol + 8 * g);
#pragma unroll
                            for (int c = 0; c < 4; ++c) x[4 * g + c] = __builtin_fmaf(-mu, cs[c], x[4 * g + c]) * rs;
                        }
                    } else if (norm_rms) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) x[r] *= rs;
                    }
                    if (has_bias) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 bv = *(const f32x4*)(p.bias + ncol + 8 * g);
#pragma unroll
                            for (int c = 0; c < 4; ++c) x[4 * g + c] += bv[c];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (ACT == ACT_QGELU) x[r] = quick_gelu_f(x[r]);
                        if (ACT == ACT_GELU) x[r] = gelu_erf_f(x[r]);
                        if (ACT == ACT_SILU) x[r] = silu_f(x[r]);
                        if (ACT == ACT_GELU_TANH) x[r] = gelu_tanh_f(x[r]);
                    }
                }
                if (has_res) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32x2 rv = *(const u32x2*)(rrow + ocol + 8 * g);
                        x[4 * g + 0] += e_lo(rv[0]);
                        x[4 * g + 1] += e_hi(rv[0]);
                        x[4 * g + 2] += e_lo(rv[1]);
                        x[4 * g + 3] += e_hi(rv[1]);
                    }
                }
                uint32_t pk[8];                                              // pk[2 g + h] = columns 8 g + 4 hi + 2 h, + 1 (bf16 pair)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pk[2 * g] = pack2bf(x[4 * g], x[4 * g + 1]);
                    pk[2 * g + 1] = pack2bf(x[4 * g + 2], x[4 * g + 3]);
                }
                // register groups (g, g + 1), g = 0, 2: afterwards lanes 0-31 hold columns 8 g .. 8 g + 7 (own group g | the upper
                // half-wave's group g), lanes 32-63 columns 8 (g + 1) .. + 7 (the lower half-wave's group g + 1 | own).  Inline asm, not
                // __builtin_amdgcn_permlane32_swap: inside this function hipcc (ROCm 7.2) folded the builtin's SECOND result into a
                // copy of the first (seen in the .s; wrong columns 4..7 of every 8 on hardware).  s_nop 1 = the two wait states a VALU
                // write of an operand needs before the swap reads it (guide T21).
                VL2_PERMLANE32_SWAP_8(pk);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const u32x4 packed = {pk[4 * pr], pk[4 * pr + 1], pk[4 * pr + 2], pk[4 * pr + 3]};
                    const int oc8 = (SWIGLU ? (n_w0 >> 1) + cb * 32 : n_w0 + nj * 32) + 16 * pr + 8 * hi;      // first of the 8 columns
                    if (live) *(u32x4*)(crow + oc8) = packed;
                    if (has_stats) {                                         // statistics of the row AS STORED, per 8-column group
                        float rf[8], ss = 0.f, qq = 0.f;
                        unpack8(packed, rf);
#pragma unroll
                        for (int j = 0; j < 8; ++j) { ss += rf[j]; qq = __builtin_fmaf(rf[j], rf[j], qq); }
                        s8[half * 2 + pr] = ss;
                        q8[half * 2 + pr] = qq;
                    }
                }
            }
            if (has_stats) {
                // the 64-column block's eight 8-column groups: this lane holds groups 2 i + hi (i = 0..3), its half-wave partner the
                // other four.  gemm_store_patch folds them with octet_sum = ((s0 + s7) + (s1 + s6)) + ((s2 + s5) + (s3 + s4)):
                // the same tree here (fp32 addition is commutative, so the operand order inside a pair does not matter).
                float ts[4], tq[4];                                          // t_g = s_g + s_{7-g} for the own groups g = 2 i + hi
#pragma unroll
                for (int i = 0; i < 4; ++i) {                                // 7 - (2 i + hi) = 2 (3 - i) + (1 - hi): the partner's slot 3 - i
                    ts[i] = s8[i] + __shfl_xor(s8[3 - i], 32);
                    tq[i] = q8[i] + __shfl_xor(q8[3 - i], 32);
                }
                // lane hi = 0: ts = (t0, t2, t4, t6) with t4 = t3, t6 = t1  ->  (t0 + t1) + (t2 + t3) = (ts0 + ts3) + (ts1 + ts2)
                const float tot_s = (ts[0] + ts[3]) + (ts[1] + ts[2]), tot_q = (tq[0] + tq[3]) + (tq[1] + tq[2]);
                if (live && hi == 0) {
                    gemm_stat_put(p, p.stats_out + ((size_t)m * p.stats_out_np + ((n_w0 + cb * 64) >> 6)) * 2, tot_s, tot_q);
                }
            }
        }
    }
}

#define VL2_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// SPLITK: gridDim.y workgroups share one output tile, each owning a contiguous 1/gridDim.y of the K-tiles (the launcher
// picks a divisor).  Used when the plain grid would leave most CUs idle (small M: the per-rank shapes of the frame-sharded
// encoder, the Conv3d-as-GEMM with K = 32768): a workgroup streams its operands at a fixed ~50-65 GB/s through LDS-DMA, so
// a lone tile's latency is K/64 x 0.64 us whatever the rest of the chip does.  Every workgroup publishes its fp32
// accumulators (sk_ws, [tile][split][64][256] floats, lane-coalesced), then takes a ticket on the tile's counter
// (sk_flags[tile]); the LAST arriver re-reads all partials in split order (deterministic sum, whoever is last), re-arms
// the counter for the next launch and runs the normal epilogue.  Hand-off = guide G16: drain, barrier, one lane
// agent-scope release + drain + relaxed ticket; the reducer: agent-scope acquire, barrier, plain loads.
template <int ACT, bool SWIGLU, bool OUT_F32, bool GATHER, bool REMAP = false, bool SPLITK = false>
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
    f32x2 rst = {0.f, 1.f};                                    // (mean, rstd) of A row m0 + tid (norm-carrying GEMMs)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nsplit = SPLITK ? (int)gridDim.y : 1;
    const int nt = p.K / GEMM_BK / nsplit;                  // K-tiles of this workgroup
    const int kt0 = SPLITK ? (int)blockIdx.y * nt : 0;      // first K-tile of this workgroup
    const int frow = lane & 31, fchk = lane >> 5;

    if constexpr (!GATHER) {
        // ---- main loop, issue-lean form.  Measured (profiles/r01_gemm_experiments.md): with two waves per SIMD the kernel is
        // bound by instruction ISSUE, not by the MFMA pipe or LDS -- so the loop carries no per-iteration VALU:
        //   * LDS-DMA as `buffer_load_dwordx4 ... offen lds`: loop-invariant 32-bit VGPR byte offsets, the K position in
        //     the SGPR soffset, the LDS slot in M0 (one s_add + one VMEM instruction per 1 KiB piece);
        //   * fragment ds_read_b128 addresses = 8 loop-invariant VGPR bases + immediate offsets (K loop unrolled by the
        //     two LDS buffers so the buffer offset is an immediate).
        const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
        const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
        unsigned a_vo[4];                           // per-thread byte offsets of the 4 A pieces (rows clamp at M-1)
        unsigned w_vo;                              // W piece 0; piece i is 32 rows further: + i * 64 * ldw bytes (uniform)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = ((i * 4 + wave) << 6) + lane;
            const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
            int am = m0 + 2 * R + (sx >> 3);
            am = am < p.M ? am : p.M - 1;
            a_vo[i] = ((unsigned)am * (unsigned)p.lda + (sx & 7) * 8) * 2;
            if (i == 0) w_vo = ((unsigned)(n0 + 2 * R + (sx >> 3)) * (unsigned)p.ldw + (sx & 7) * 8) * 2;
        }
        const unsigned w_step = 64u * (unsigned)p.ldw;                    // bytes between W pieces (32 rows)
        unsigned a_rd[4], b_rd[4];                   // ds_read bases per k-step (tile i = 1 is +4096 B)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a_rd[ks] = gemm_lds_off(wm * 64 + frow, ks * 2 + fchk);
            b_rd[ks] = 16384 + gemm_lds_off(wn * 64 + frow, ks * 2 + fchk);
        }
        auto stage = [&](unsigned lds_buf, int kt) {
            const unsigned kb = (unsigned)(kt0 + kt) * (GEMM_BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + ((i * 4 + wave) << 10)),
                                                         16, a_vo[i], kb, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + 16384 + ((i * 4 + wave) << 10)),
                                                         16, w_vo, kb + i * w_step, 0, 0);
        };
        auto compute = [&](unsigned lds_buf) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *(const bf16x8*)(vl2_smem + a_rd[ks] + (lds_buf + i * 4096));
                    bfr[i] = *(const bf16x8*)(vl2_smem + b_rd[ks] + (lds_buf + i * 4096));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = VL2_MFMA32(af[i], bfr[j], acc[i][j]);
            }
        };
        stage(0, 0);
        // the rows' statistics are fetched (and, without a finalize table, reduced) BEHIND the first slab's LDS-DMA: their latency
        // rides on the ring fill.  VL2_PIN2 keeps the optimiser from sinking the pure loads to their first use in the epilogue.
        rst = gemm_row_stats(p, m0, tid, GEMM_BM);
        VL2_PIN2(rst[0], rst[1]);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 <= nt; kt += 2) {              // two K-tiles per trip (one per LDS buffer), no exit inside the body
            stage(32768, kt + 1);
            compute(0);
            __syncthreads();
            if (kt + 2 < nt) stage(0, kt + 2);
            compute(32768);
            __syncthreads();
        }
        if (kt < nt) {                               // odd K-tile count: the last tile sits in buffer 0
            compute(0);
            __syncthreads();
        }
    } else {
        // ---- gathered-A form (Conv3d taps): the A row of (K segment, m) comes from an index table.  Same issue-lean loop;
        // the 4 per-thread A offsets are reloaded when the staged K-tile enters a new segment (every seg_k/64 tiles), and a
        // missing tap (index < 0) is an out-of-range buffer offset, which the hardware reads as zeros (no zero page needed).
        const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
        const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
        int g_row[4];
        unsigned g_chk[4], a_vo[4], w_vo = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = ((i * 4 + wave) << 6) + lane;
            const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
            int am = m0 + 2 * R + (sx >> 3);
            g_row[i] = am < p.M ? am : p.M - 1;
            g_chk[i] = (sx & 7) * 16;
            a_vo[i] = 0x80000000u;
            if (i == 0) w_vo = ((unsigned)(n0 + 2 * R + (sx >> 3)) * (unsigned)p.ldw + (sx & 7) * 8) * 2;
        }
        const unsigned w_step = 64u * (unsigned)p.ldw;
        const int tps = p.seg_k / GEMM_BK;            // K-tiles per segment
        unsigned a_rd[4], b_rd[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a_rd[ks] = gemm_lds_off(wm * 64 + frow, ks * 2 + fchk);
            b_rd[ks] = 16384 + gemm_lds_off(wn * 64 + frow, ks * 2 + fchk);
        }
        auto stage = [&](unsigned lds_buf, int ktl) {
            const int kt = kt0 + ktl;
            const int seg = kt / tps, kl = kt - seg * tps;
            if (kl == 0 || ktl == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = p.a_idx[(size_t)seg * p.idx_ld + g_row[i]];
                    a_vo[i] = r < 0 ? 0x80000000u : (unsigned)r * (unsigned)p.lda * 2u + g_chk[i];
                }
            }
            const unsigned ka = (unsigned)kl * (GEMM_BK * 2), kw = (unsigned)kt * (GEMM_BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + ((i * 4 + wave) << 10)),
                                                         16, a_vo[i], ka, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + 16384 + ((i * 4 + wave) << 10)),
                                                         16, w_vo, kw + i * w_step, 0, 0);
        };
        auto compute = [&](unsigned lds_buf) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *(const bf16x8*)(vl2_smem + a_rd[ks] + (lds_buf + i * 4096));
                    bfr[i] = *(const bf16x8*)(vl2_smem + b_rd[ks] + (lds_buf + i * 4096));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = VL2_MFMA32(af[i], bfr[j], acc[i][j]);
            }
        };
        stage(0, 0);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 <= nt; kt += 2) {
            stage(32768, kt + 1);
            compute(0);
            __syncthreads();
            if (kt + 2 < nt) stage(0, kt + 2);
            compute(32768);
            __syncthreads();
        }
        if (kt < nt) {
            compute(0);
            __syncthreads();
        }
    }

    if constexpr (SPLITK) {
        float* ws = p.sk_ws + ((size_t)t * nsplit + blockIdx.y) * (64 * 256);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[((i * 2 + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        VL2_DRAIN_VMEM();
        __syncthreads();
        int* is_last = (int*)(vl2_smem + GEMM_LDS_BYTES - 16);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            VL2_DRAIN_VMEM();
            const int ticket = __hip_atomic_fetch_add(p.sk_flags + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == nsplit - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(p.sk_flags + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
            }
            *is_last = ticket == nsplit - 1;
        }
        __syncthreads();
        if (!*is_last) return;
        const float* w0 = p.sk_ws + (size_t)t * nsplit * (64 * 256);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = w0[((i * 2 + j) * 16 + r) * 256 + tid];
        for (int sp = 1; sp < nsplit; ++sp) {
            const float* wsp = w0 + (size_t)sp * (64 * 256);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += wsp[((i * 2 + j) * 16 + r) * 256 + tid];
        }
        __syncthreads();
    }

    // ---- epilogue: acc (col = lane&31, rows (r&3)+8(r>>2)+4(lane>>5)) -> fp32 LDS patch [32][68] per wave -> rows
    float* ep = (float*)vl2_smem + wave * (32 * 68);
    float* rowtab = (float*)vl2_smem + 4 * (32 * 68);          // behind the four wave patches
    gemm_park_row_stats(p, rowtab, rst, tid, GEMM_BM);
    __syncthreads();                       // the row table is visible; from here on every wave works on its OWN patch: the LDS
    // operations of one wave execute in issue order, so the patch writes of pass k+1 cannot overtake the reads of pass k and no
    // workgroup barrier separates the passes -- the waves' store chains (LDS -> residual load -> store) overlap freely
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ep[row * 68 + ni * 32 + (lane & 31)] = acc[mi][ni][r];
            }
        __builtin_amdgcn_wave_barrier();    // no instruction on the hardware (a wave's lanes run in lockstep and its LDS operations
        gemm_store_patch<ACT, SWIGLU, OUT_F32, REMAP>(p, ep, m0 + wm * 64 + mi * 32, n0 + wn * 64, lane, rowtab, wm * 64 + mi * 32);
        __builtin_amdgcn_wave_barrier();    // complete in order); pins the write / read order for the compiler and the CPU emulator
    }
    if constexpr (!SWIGLU && !OUT_F32 && !REMAP && !SPLITK) gemm_rows_ticket<256>(p, tm, m0, GEMM_BM, tid);
}

// ------------------------------------------------------------------------------------------------------------------
// gemm_sk: stream-K form of gemm_bf16_kernel for grids that quantise badly on 256 CUs x 2 workgroups (e.g. the 416 / 584 /
// 624 tiles of the M=1621 prefill and N=1024 ViT GEMMs).  A fixed grid of persistent workgroups splits the flattened
// (tile, K-tile) iteration space evenly; a workgroup's range is contiguous, so it holds at most one tile TAIL (at the start
// of its range -> it stores its fp32 accumulators and raises a flag at once) and at most one tile HEAD (at the end of its
// range -> it adds the partials of the following workgroups, then runs the normal epilogue).  Hand-off = the release /
// acquire recipe of the guide (G16): plain 16-B stores, every wave drains vmcnt, barrier, one lane: agent-scope release
// fence + drain + relaxed agent flag store; consumer: one lane polls relaxed, agent-scope acquire, barrier, plain loads.
// The waiter always waits on HIGHER logical workgroup ids whose contribution is the FIRST thing they compute, so there is
// no wait chain; every spin is bounded (a timeout leaves the tile unreduced and sets flag slot [grid]).

template <int ACT, bool SWIGLU, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_sk_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, G);                      // logical id: every XCD owns a contiguous iteration range
    const int nt = p.K / GEMM_BK;
    const int total = p.tiles_m * p.tiles_n * nt;
    int it = bid * p.sk_per;
    int it_end = it + p.sk_per;
    it_end = it_end < total ? it_end : total;
    const int frow = lane & 31, fchk = lane >> 5;
    float* ep = (float*)vl2_smem + wave * (32 * 68);

    while (it < it_end) {
        const int tile = it / nt, k0 = it - tile * nt;
        int k1 = k0 + (it_end - it);
        k1 = k1 < nt ? k1 : nt;
        const int grp_sz = 8 * p.tiles_n;
        const int first_m = (tile / grp_sz) * 8;
        const int gm = (p.tiles_m - first_m) < 8 ? (p.tiles_m - first_m) : 8;
        const int tm = first_m + (tile % grp_sz) % gm, tn = (tile % grp_sz) / gm;
        const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

        unsigned a_off[4], b_off[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = ((i * 4 + wave) << 6) + lane;
            const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
            int am = m0 + 2 * R + (sx >> 3);
            am = am < p.M ? am : p.M - 1;
            a_off[i] = (unsigned)am * (unsigned)p.lda + (sx & 7) * 8;
            b_off[i] = (unsigned)(n0 + 2 * R + (sx >> 3)) * (unsigned)p.ldw + (sx & 7) * 8;
        }
        auto stage = [&](int buf, int kt) {
            unsigned char* As = vl2_smem + buf * 32768;
            unsigned char* Bs = As + 16384;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(p.A + a_off[i] + kt * GEMM_BK, As + ((i * 4 + wave) << 10));
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(p.W + b_off[i] + kt * GEMM_BK, Bs + ((i * 4 + wave) << 10));
        };

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        stage(0, k0);
        __syncthreads();
        for (int kt = k0; kt < k1; ++kt) {
            const int buf = (kt - k0) & 1;
            if (kt + 1 < k1) stage(buf ^ 1, kt + 1);
            const unsigned char* As = vl2_smem + buf * 32768;
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
                        acc[i][j] = VL2_MFMA32(af[i], bfr[j], acc[i][j]);
            }
            __syncthreads();
        }

        if (k0 != 0) {
            // ---- tile tail: publish the partial accumulators (thread-private layout: element e of thread tid at [e][tid])
            float* ws = p.sk_ws + (size_t)bid * (64 * 256);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ws[((i * 2 + j) * 16 + r) * 256 + tid] = acc[i][j][r];
            VL2_DRAIN_VMEM();
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                VL2_DRAIN_VMEM();
                __hip_atomic_store(p.sk_flags + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (k1 != nt) {
                // ---- tile head: add the partials of the workgroups that own the rest of this tile's K range
                int rem = nt - k1, c = bid + 1;
                while (rem > 0) {
                    if (tid == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(p.sk_flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1 << 22)) { __hip_atomic_store(p.sk_flags + G, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const float* ws = p.sk_ws + (size_t)c * (64 * 256);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] += ws[((i * 2 + j) * 16 + r) * 256 + tid];
                    rem -= p.sk_per < rem ? p.sk_per : rem;
                    ++c;
                }
            }
            // ---- epilogue (as gemm_bf16_kernel)
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
        it += k1 - k0;
    }
}

// s_waitcnt with only vmcnt / only lgkmcnt active (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
#define VL2_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))
#define VL2_WAIT_LGKMCNT0() __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14))

// ------------------------------------------------------------------------------------------------------------------
// gemm4 "ping-pong": 256 x 256 block tile, K streamed in 32-deep slabs through a 4-stage LDS ring.
// PMC on the 128x128 kernel shows each wave ~52 % of its cycles stalled on MFMA issue behind its SIMD partner and ~27 %
// parked at waitcnt/barrier: with two identical in-order waves per SIMD, a wave stuck behind the partner's MFMA cannot issue
// its own ds_reads / LDS-DMA either.  Here the two waves of a SIMD run in anti-phase: in every phase one group issues ONLY
// MFMAs (operands already in registers) while the other issues ONLY memory work, phases separated by a raw s_barrier.
// Halves the L2 -> LDS operand stream per FLOP relative to the 128-wide kernels (7.6 B/kFLOP), which is what bounds them
// (profiles/r01_gemm_experiments.md).  8 waves = two groups of 4 (waves w, w+4 share a SIMD); group g owns rows
// [128g, 128g+128) x all 256 columns, wave tile 64 x 128 = 2 x 4 MFMA 32x32x16 accumulators (128 VGPRs).
// A phase = one K-slab of 32: the MFMA group issues 16 MFMAs from 12 register fragments while the other group reads its
// next 12 fragments and issues its 4 LDS-DMA (its 128 rows of A and of W) for the slab three ahead.
//   group g: LOAD(t) at phase 2t+g, MFMA(t) one phase later.  Stage (t+3)%4 last held slab t-1 (read in phases 2t-2,
//   2t-1), so DMA(t+3) may be issued from phase 2t on; it is retired by a COUNTED vmcnt(8) two LOAD phases later (the
//   two newer slabs stay in flight), i.e. every LDS-DMA has ~5 phases of flight time.
// LDS image per operand slab: 256 rows x 64 B, 4 rows per 256-B bank row, chunk c of row r stored at c ^ ((r>>2)&3).
#define GEMM4_BM 256
#define GEMM4_BN 256
#define GEMM4_BK 32
#define GEMM4_STAGE 32768
#define GEMM4_LDS_BYTES (4 * GEMM4_STAGE)
// phase boundary: the compiler may not move MFMAs (register-only, so not ordered by the barrier itself) across it
#define VL2_PHASE_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ int gemm4_lds_off(int row, int chunk) {
    return (((row >> 2) << 4) + ((row & 3) << 2) + (chunk ^ ((row >> 2) & 3))) << 4;
}

// bid / nwg: this workgroup's index and the workgroup count of the tile set it belongs to (the whole grid for
// gemm4_bf16_kernel; the big-tile part of gemm_mix_bf16_kernel)
// TR: accumulate C^T (MFMA operands swapped) and store through gemm_store_tr (no LDS in the epilogue); bf16 output only
// BM = 256: group g owns rows [128g, +128), its four waves 2 x 2 (wave tile 64 x 128).
// BM = 192: the same kernel on a 192 x 256 tile for GEMMs that leave a 256-row grid well short of one round (ViT out_proj / fc2 at
//   16 frames: 37 x 4 = 148 tiles of 256 rows on 256 CUs, 49 x 4 = 196 tiles of 192 rows): group g owns rows [96g, +96), its four
//   waves side by side (wave tile 96 x 64 = 3 x 2 accumulators, 5 fragment reads per 6 MFMAs), the A slab is 12 LDS-DMA pieces
//   (waves 0, 1 of a group issue two, waves 2, 3 one: counted vmcnt per wave).  Same slabs, same k order -> same bits per element.
// FP8: A and W rows hold e4m3fn bytes (a slab = 64 k per row in the same 64 bytes); one v_mfma_f32_32x32x64_f8f6f4 per accumulator block
//   and slab from the two fragments the 16-bit form feeds to two MFMAs (dev_common.h VL2_MFMA32_F8); epilogue = gemm_store_patch with the
//   row / column scales (GemmArgs.row_norm, col_scale).
// WEAVE4 (round 5): the LDS-DMA pieces of slab t+3 are issued from the wave's own MFMA(t) phase, one behind every fourth MFMA, instead of from its
//   LOAD(t) phase.  The load phase of a group must fit under its partner's matrix phase (16 MFMAs = 512 cycles) and does not: four LDS-DMA pieces cost
//   60-100 cycles of issue each (MI355X_MICROARCH.md) next to 12 fragment reads and two barriers.  Unlike the 3-stage kernels' weave (one phase of flight
//   for the woven pieces: lost on cold operands) the 4-stage ring leaves them 2.5 slab times.  The slot of slab t+3 is slab t-1's, dead for both groups
//   by MFMA(t).  Same slabs, same k order -> same bits.
template <int ACT, bool SWIGLU, bool OUT_F32, bool TR = false, int EF = -1, int BM = 256, bool FP8 = false, bool WEAVE4 = false>
__device__ __forceinline__ void gemm4_body(const GemmArgs& p, int bid, int nwg) {
    static_assert(!(TR && OUT_F32), "gemm_store_tr writes bf16");
    static_assert(!(FP8 && TR), "the fp8 form stores through gemm_store_patch");
    static_assert(BM == 256 || BM == 192 || BM == 160, "gemm4: 256-, 192- or 160-row tiles");
    static_assert(BM != 160 || (!TR && !FP8 && !WEAVE4), "gemm4: the 160-row tile is built with the LDS epilogue and the load-phase issue");
    constexpr int GR = BM == 256 ? 128 : 96;                       // rows of wave group 0 (= the row offset of group 1)
    constexpr int MI = BM == 256 ? 2 : 3, NJ = BM == 256 ? 4 : 2;  // 32 x 32 accumulator blocks of a wave
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wrow = BM == 256 ? grp * 128 + (w4 >> 1) * 64 : grp * 96;     // first row / column of the wave tile inside the block tile
    const int wcol = BM == 256 ? (w4 & 1) * 128 : w4 * 64;

    // BM = 160 (round 6): the 192-row tile WITHOUT the third row block of group 1 -- group 0 owns rows [0, 96) (3 x 2 blocks per wave), group 1 rows
    //   [96, 160) (2 x 2 blocks per wave: its third block is never loaded, multiplied or stored).  For GEMMs whose 192-row grid is one badly filled
    //   round: ViT out_proj / fc2 at 16 frames are 49 x 4 = 196 tiles of 192 rows on 256 CUs (77 %), 58 x 4 = 232 tiles of 160 rows (91 %), and a
    //   one-round grid takes one tile time whatever its fill.  The two groups' matrix phases are 12 and 8 MFMAs per slab (a SIMD carries one wave
    //   of each: 20 per slab instead of 24).  Same slabs, same k order -> same bits per element.
    const bool blk2 = !(BM == 160 && grp == 1);                    // this wave has its third row block (wave-uniform)
    const int t0 = xcd_remap(bid, nwg);
    const int grp_sz = 4 * p.tiles_n;                              // 4 tile-rows (1024 rows of A) per raster group
    const int first_m = (t0 / grp_sz) * 4;
    const int gm = (p.tiles_m - first_m) < 4 ? (p.tiles_m - first_m) : 4;
    const int tm = first_m + (t0 % grp_sz) % gm, tn = (t0 % grp_sz) / gm;
    const int m0 = tm * BM, n0 = tn * GEMM4_BN;
    f32x2 rst = {0.f, 1.f}, rowst[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rowst[i] = f32x2{0.f, 1.f};

    // this wave's LDS-DMA parts of a slab: its share of the group's A rows [GR*grp, +GR) and 2 x W rows [128*grp, +128), issue-lean
    // form: `buffer_load_dwordx4 ... offen lds` with loop-invariant VGPR byte offsets, K position in the SGPR soffset
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    constexpr int APG = GR / 16;                                   // A pieces (16 rows each) of a group: 8 or 6
    const bool two_a = BM == 256 || (w4 < 2 && blk2);              // this wave's second A piece exists (wave-uniform; 160 rows: group 1 has four pieces)
    unsigned a_vo[2], w_vo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int aslot = ((grp * APG + i * 4 + w4) << 6) + lane, wslot = grp * 512 + ((i * 4 + w4) << 6) + lane;
        const int Ra = aslot >> 4, spa = aslot & 15, Rw = wslot >> 4, spw = wslot & 15;
        const int rowa = 4 * Ra + (spa >> 2), chka = (spa & 3) ^ (Ra & 3);
        const int roww = 4 * Rw + (spw >> 2), chkw = (spw & 3) ^ (Rw & 3);
        int am = m0 + rowa;
        am = am < p.M ? am : p.M - 1;
        a_vo[i] = ((unsigned)am * (unsigned)p.lda + chka * 8) * 2;
        w_vo[i] = ((unsigned)(n0 + roww) * (unsigned)p.ldw + chkw * 8) * 2;
    }
    auto issue_dma = [&](int t) {
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE, kb = (unsigned)t * (GEMM4_BK * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * APG + w4) << 10)),
                                                 16, a_vo[0], kb, 0, 0);
        if (two_a)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * APG + 4 + w4) << 10)),
                                                     16, a_vo[1], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((grp * 8 + i * 4 + w4) << 10)),
                                                     16, w_vo[i], kb, 0, 0);
    };
    auto issue_piece = [&](int t, int tk, int i) {    // piece i of this wave's share of slab tk, into the ring slot of slab t: 0, 1 = A (1 only if two_a), 2, 3 = W
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE, kb = (unsigned)tk * (GEMM4_BK * 2);
        if (i == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * APG + w4) << 10)), 16, a_vo[0], kb, 0, 0);
        else if (i == 1) {
            if (two_a)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * APG + 4 + w4) << 10)), 16, a_vo[1], kb, 0, 0);
        } else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((grp * 8 + (i - 2) * 4 + w4) << 10)),
                                                     16, w_vo[i - 2], kb, 0, 0);
    };
    // counted waits: `slabs` newer slabs of this wave's LDS-DMA stay in flight (4 or 3 pieces per slab)
    auto wait_dma = [&](int slabs) {
        if (slabs >= 2) { if (two_a) VL2_WAIT_VMCNT(8); else VL2_WAIT_VMCNT(6); }
        else if (slabs == 1) { if (two_a) VL2_WAIT_VMCNT(4); else VL2_WAIT_VMCNT(3); }
        else VL2_WAIT_VMCNT(0);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][MI], fb[2][NJ];                   // [ks][tile]

    const int nt = p.K / GEMM4_BK;
    const int frow = lane & 31, fchk = lane >> 5;
    const int arow = wrow + frow, brow = wcol + frow;
    unsigned a_rd[2], b_rd[2];                     // fragment read bases per k-step; tile i / j is +2048 B (32 rows)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_rd[ks] = gemm4_lds_off(arow, ks * 2 + fchk);
        b_rd[ks] = 16384 + gemm4_lds_off(brow, ks * 2 + fchk);
    }
    issue_dma(0);
    if (nt > 1) issue_dma(1);
    if (nt > 2) issue_dma(2);
    // the rows' statistics are fetched (and, without a finalize table, reduced) BEHIND the first three slabs' LDS-DMA: their latency
    // rides on the ring fill (ahead of the DMAs it was exposed: the in-kernel reduction measured +0.26 ms per ViT pass that way).
    // The pins keep the optimiser from sinking the pure loads to their first use in the epilogue.
    // NOTE: vmcnt retires in order and hipcc's wait for a pin behind this branchy prologue is `s_waitcnt vmcnt(0)`, so the
    // first MFMA phase of a norm-carrying GEMM starts when all three slabs have landed, and the counted wait_dma(2) below only matters
    // for GEMMs without a norm.  Three other forms were compiled and read in the ISA (round 4, profiles/r04_experiments.md section 10):
    // loads between slab 0 and slabs 1 / 2 with the pin behind them, or behind the counted wait -> still vmcnt(0) (the waitcnt pass merges
    // the branches' pending counts conservatively); a volatile load -> the memory legaliser waits for it at once, which would serialise
    // slab 0 -> statistics -> slabs 1, 2.  An untracked inline-asm load with manual counts is the form left; not taken: the register
    // allocator may copy its result before the data arrive.  The counted vmcnt values everywhere assume no other VMEM operation in flight.
    if constexpr (!TR) { rst = gemm_row_stats(p, m0, tid, BM); VL2_PIN2(rst[0], rst[1]); }
    else {
        gemm_tr_row_stats<MI>(p, m0 + wrow, lane, rowst);
#pragma unroll
        for (int i = 0; i < MI; ++i) VL2_PIN2(rowst[i][0], rowst[i][1]);
    }
    // (woven form at nt = 2: its in-loop wait keeps ONE newer slab in flight, which would be slab 1 itself -- wait for both here)
    wait_dma(nt > 2 ? 2 : (nt > 1 && !WEAVE4) ? 1 : 0);
    VL2_PHASE_BARRIER();

    if (grp == 1) VL2_PHASE_BARRIER();
    for (int t = 0; t < nt; ++t) {
        // ---------------- LOAD(t): memory work only
        if (!WEAVE4 && t + 3 < nt) issue_dma(t + 3);
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned ab = a_rd[ks] + st, bb = b_rd[ks] + st;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                if (i < 2 || blk2) fa[ks][i] = *(const bf16x8*)(vl2_smem + ab + i * 2048);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[ks][j] = *(const bf16x8*)(vl2_smem + bb + j * 2048);
        }
        // slab t+1 must have landed before the barrier below; the (up to) two newer slabs stay in flight (woven form: slab t+3 is not issued yet)
        // (woven form: ONE newer slab is in flight in every iteration -- past the end of K the matrix phases re-issue the last slab into the dead
        //  slot, so the loop has no tail case and no branch; the stray pieces are drained behind the loop)
        if constexpr (WEAVE4) wait_dma(1); else wait_dma(nt - 2 - t);                      // slabs issued after t+1
        VL2_WAIT_LGKMCNT0();
        VL2_PHASE_BARRIER();
        // ---------------- MFMA(t): matrix work only
        if constexpr (FP8) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = VL2_MFMA32_F8(fa[0][i], fa[1][i], fb[0][j], fb[1][j], acc[i][j]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (BM == 160 && i == 2 && !blk2) continue;
                        acc[i][j] = TR ? VL2_MFMA32(fb[ks][j], fa[ks][i], acc[i][j])
                                       : VL2_MFMA32(fa[ks][i], fb[ks][j], acc[i][j]);
                        if constexpr (WEAVE4) {
                            constexpr int NM = 2 * MI * NJ;                  // MFMAs of the phase: 16 (256 rows) or 12 (192 rows)
                            const int idx = (ks * MI + i) * NJ + j;
                            if (idx % (NM / 4) == 1) {                       // behind MFMA 1, 5, 9, 13 (1, 4, 7, 10): one piece each
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(t + 3, t + 3 < nt ? t + 3 : nt - 1, idx / (NM / 4));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
        }
        VL2_PHASE_BARRIER();
    }
    if constexpr (WEAVE4) VL2_WAIT_VMCNT(0);                      // the stray pieces of the last three matrix phases have landed (the epilogue reuses the ring)
    if (grp == 0) VL2_PHASE_BARRIER();

    if constexpr (TR) {       // ---- register-resident epilogue: the accumulators hold C^T, rows are lane-local
        gemm_store_tr<ACT, SWIGLU, MI, NJ, EF>(p, acc, m0 + wrow, n0 + wcol, lane, rowst);
        if constexpr (!SWIGLU && !FP8) gemm_rows_ticket<512>(p, tm, m0, BM, tid);
        return;
    }
    // ---- epilogue: 32 x 64 patches (BM = 256: 2 row blocks x 2 column halves per wave; BM = 192: 3 row blocks)
    float* ep = (float*)vl2_smem + wave * (32 * 68);
    float* rowtab = (float*)vl2_smem + 8 * (32 * 68);
    gemm_park_row_stats(p, rowtab, rst, tid, BM);
    __syncthreads();                       // row table visible; the passes below are wave-private (see gemm_bf16_kernel)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nh = 0; nh < NJ / 2; ++nh) {
            if (BM == 160 && mi == 2 && !blk2) continue;           // (rows 160 .. 191 of the 192-row layout belong to the next tile)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    ep[row * 68 + ni * 32 + (lane & 31)] = acc[mi][nh * 2 + ni][r];
                }
            __builtin_amdgcn_wave_barrier();
            gemm_store_patch<ACT, SWIGLU, OUT_F32>(p, ep, m0 + wrow + mi * 32, n0 + wcol + nh * 64, lane, rowtab, wrow + mi * 32);
            __builtin_amdgcn_wave_barrier();
        }
    if constexpr (!SWIGLU && !OUT_F32 && !FP8) gemm_rows_ticket<512>(p, tm, m0, BM, tid);
}
template <int ACT, bool SWIGLU, bool OUT_F32, bool TR = false, int EF = -1, int BM = 256, bool WEAVE4 = false>
__global__ __launch_bounds__(512, 2) void gemm4_bf16_kernel(GemmArgs p) {
    gemm4_body<ACT, SWIGLU, OUT_F32, TR, EF, BM, false, WEAVE4>(p, blockIdx.x, gridDim.x);
}
template <int ACT, bool SWIGLU, bool OUT_F32, int BM>
__global__ __launch_bounds__(512, 2) void gemm4_fp8_kernel(GemmArgs p) {
    gemm4_body<ACT, SWIGLU, OUT_F32, false, -1, BM, true>(p, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------
// gemm3: 128 (M) x 256 (N) ping-pong kernel (see gemm4) for GEMMs whose M granularity matters (the M = 1621 prefill):
// the two wave groups SHARE the 128-row A tile and own one 128-column half of W each (wave tile 64 x 64).
// K-tile 64, 3-stage ring of (A 16 KiB | W_0 16 KiB | W_1 16 KiB) = 144 KiB, LOAD phase = 16 ds_read_b128 + 6 LDS-DMA
// (issue-lean form: SGPR K offset, immediate read offsets), counted vmcnt(6), raw s_barrier.
#define GEMM3_BM 128
#define GEMM3_BN 256
#define GEMM3_STAGE 49152
#define GEMM3_LDS_BYTES (3 * GEMM3_STAGE)

// WEAVE (lab, variant 5): the six LDS-DMA pieces of K-tile t + 2 are issued from the wave's own MFMA(t) phase, one behind every second
// MFMA, instead of from its LOAD(t) phase (see k_gemm7.h gemm7_loop); they are retired by vmcnt(0) at the end of the next LOAD phase.
// FP8: see gemm4_body -- two v_mfma_f32_32x32x64_f8f6f4 per accumulator block and K-tile (128 k per row in the tile's 128 bytes).
template <int ACT, bool SWIGLU, bool OUT_F32, bool TR = false, int EF = -1, bool WEAVE = false, bool FP8 = false>
__device__ __forceinline__ void gemm3_body(const GemmArgs& p, int bid, int nwg) {
    static_assert(!(TR && OUT_F32), "gemm_store_tr writes bf16");
    static_assert(!(FP8 && (TR || WEAVE)), "the fp8 form: LDS epilogue, load-phase DMA issue");
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;                      // waves w and w+4 share a SIMD (measured)
    const int wm = w4 >> 1, wn = w4 & 1;

    const int t0 = xcd_remap(bid, nwg);
    const int grp_sz = 8 * p.tiles_n;
    const int first_m = (t0 / grp_sz) * 8;
    const int gm = (p.tiles_m - first_m) < 8 ? (p.tiles_m - first_m) : 8;
    const int tm = first_m + (t0 % grp_sz) % gm, tn = (t0 % grp_sz) / gm;
    const int m0 = tm * GEMM3_BM, n0 = tn * GEMM3_BN;
    f32x2 rst = {0.f, 1.f}, rowst[2] = {{0.f, 1.f}, {0.f, 1.f}};
    if constexpr (!TR) rst = gemm_row_stats(p, m0, tid, GEMM3_BM);
    else gemm_tr_row_stats<2>(p, m0 + wm * 64, lane, rowst);

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    // this wave's LDS-DMA pieces of a K-tile: 4 x W_grp (its own 128 rows; piece i is 32 rows further) + 2 x A rows [64*grp, +64)
    unsigned w_vo, a_vo[2];
    {
        const int slot = (w4 << 6) + lane;
        const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
        w_vo = ((unsigned)(n0 + grp * 128 + 2 * R + (sx >> 3)) * (unsigned)p.ldw + (sx & 7) * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = (((grp * 2 + i) * 4 + w4) << 6) + lane;
        const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
        int am = m0 + 2 * R + (sx >> 3);
        am = am < p.M ? am : p.M - 1;
        a_vo[i] = ((unsigned)am * (unsigned)p.lda + (sx & 7) * 8) * 2;
    }
    const unsigned w_step = 64u * (unsigned)p.ldw;
    auto issue_piece = [&](int kt, int i) {                     // piece i of this wave's share of K-tile kt: 0..3 = W, 4..5 = A
        const unsigned st = (unsigned)(kt % 3) * GEMM3_STAGE, kb = (unsigned)kt * (GEMM_BK * 2);
        if (i < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + grp * 16384 + ((i * 4 + w4) << 10)),
                                                     16, w_vo, kb + i * w_step, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + (((grp * 2 + i - 4) * 4 + w4) << 10)),
                                                     16, a_vo[i - 4], kb, 0, 0);
    };
    auto issue_dma = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(kt, i);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[4][2], fb[4][2];

    const int nt = p.K / GEMM_BK;
    const int frow = lane & 31, fchk = lane >> 5;
    unsigned a_rd[4], b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_rd[ks] = gemm_lds_off(wm * 64 + frow, ks * 2 + fchk);
        b_rd[ks] = 16384 + grp * 16384 + gemm_lds_off(wn * 64 + frow, ks * 2 + fchk);
    }
    issue_dma(0);
    if (nt > 1) issue_dma(1);
    if (nt > 1) VL2_WAIT_VMCNT(6); else VL2_WAIT_VMCNT(0);
    VL2_PHASE_BARRIER();

    if (grp == 1) VL2_PHASE_BARRIER();
    for (int t = 0; t < nt; ++t) {
        // ---------------- LOAD(t)
        const bool more = t + 2 < nt;
        if (!WEAVE && more) issue_dma(t + 2);
        const unsigned st = (unsigned)(t % 3) * GEMM3_STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned ab = a_rd[ks] + st, bb = b_rd[ks] + st;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[ks][i] = *(const bf16x8*)(vl2_smem + ab + i * 4096);
                fb[ks][i] = *(const bf16x8*)(vl2_smem + bb + i * 4096);
            }
        }
        if (!WEAVE && more) VL2_WAIT_VMCNT(6); else VL2_WAIT_VMCNT(0);      // tile t+1 landed; the 6 just issued stay in flight
        VL2_WAIT_LGKMCNT0();
        VL2_PHASE_BARRIER();
        // ---------------- MFMA(t)
        if constexpr (FP8) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = VL2_MFMA32_F8(fa[2 * k2][i], fa[2 * k2 + 1][i], fb[2 * k2][j], fb[2 * k2 + 1][j], acc[i][j]);
        } else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = TR ? VL2_MFMA32(fb[ks][j], fa[ks][i], acc[i][j])
                                   : VL2_MFMA32(fa[ks][i], fb[ks][j], acc[i][j]);
                    if constexpr (WEAVE) {
                        const int idx = ks * 4 + i * 2 + j;
                        if ((idx & 1) && idx < 12) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) issue_piece(t + 2, idx >> 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
        VL2_PHASE_BARRIER();
    }
    if (grp == 0) VL2_PHASE_BARRIER();

    if constexpr (TR) {       // register-resident epilogue (the accumulators hold C^T)
        gemm_store_tr<ACT, SWIGLU, 2, 2, EF>(p, acc, m0 + wm * 64, n0 + grp * 128 + wn * 64, lane, rowst);
        if constexpr (!SWIGLU && !FP8) gemm_rows_ticket<512>(p, tm, m0, GEMM3_BM, tid);
        return;
    }
    float* ep = (float*)vl2_smem + wave * (32 * 68);
    float* rowtab = (float*)vl2_smem + 8 * (32 * 68);
    gemm_park_row_stats(p, rowtab, rst, tid, GEMM3_BM);
    __syncthreads();                       // row table visible; the passes below are wave-private (see gemm_bf16_kernel)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ep[row * 68 + ni * 32 + (lane & 31)] = acc[mi][ni][r];
            }
        __builtin_amdgcn_wave_barrier();
        gemm_store_patch<ACT, SWIGLU, OUT_F32>(p, ep, m0 + wm * 64 + mi * 32, n0 + grp * 128 + wn * 64, lane, rowtab, wm * 64 + mi * 32);
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (!SWIGLU && !OUT_F32 && !FP8) gemm_rows_ticket<512>(p, tm, m0, GEMM3_BM, tid);
}
template <int ACT, bool SWIGLU, bool OUT_F32, bool TR = false, int EF = -1, bool WEAVE = false>
__global__ __launch_bounds__(512, 2) void gemm3_bf16_kernel(GemmArgs p) {
    gemm3_body<ACT, SWIGLU, OUT_F32, TR, EF, WEAVE>(p, blockIdx.x, gridDim.x);
}
template <int ACT, bool SWIGLU, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm3_fp8_kernel(GemmArgs p) {
    gemm3_body<ACT, SWIGLU, OUT_F32, false, -1, false, true>(p, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------
// gemm_s "small-M" form: 64 x 64 block tile, 2 waves (wave w owns rows [32w, 32w+32) x 64 columns = 2 accumulators),
// BK = 64, 4-stage 64 KiB LDS ring with counted vmcnt and ONE raw barrier per K-tile.
// Why it exists: a workgroup pulls its operands through LDS-DMA at a fixed ~50-60 GB/s whatever the rest of the chip does
// (scripts/kernel_bench.py --small: a lone 128x128 tile with K = 4096 takes 41-46 us for 2 MB of operands), so a GEMM
// whose 128x128 grid has fewer tiles than there are CUs is bound by (bytes per workgroup) / 55 GB/s, not by FLOPs.  That
// is the regime of the frame-sharded encoder (2-4 frames per rank: M = 1154...2308, or 169...507 output-frame rows) and
// of short videos.  Quartering the tile puts 4x the workgroups on the idle CUs with half the bytes each.  The K order of
// the accumulation is the same as in every other kernel here (sequential 16-deep MFMA steps), so a row's result does not
// depend on which kernel -- i.e. on M -- computed it: a sharded run stays bit-identical to the single-GPU run.
#define GEMMS_BM 64
#define GEMMS_BN 64
#define GEMMS_STAGES 4
#define GEMMS_STAGE_BYTES 16384
#define GEMMS_LDS_BYTES (GEMMS_STAGES * GEMMS_STAGE_BYTES)

template <int ACT, bool GATHER>
__global__ __launch_bounds__(128, 2) void gemm_s_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // consecutive workgroups (one XCD after the remap) walk the M tiles of one W panel: the panel is fetched once per XCD
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = t % p.tiles_m, tn = t / p.tiles_m;
    const int m0 = tm * GEMMS_BM, n0 = tn * GEMMS_BN;
    const int nt = p.K / GEMM_BK;
    const f32x2 rst = gemm_row_stats(p, m0, tid, GEMMS_BM);

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    // stage image: A 64 rows x 128 B (8 pieces of 1 KiB) | W 64 rows x 128 B; wave w issues pieces 2i + w; same bank swizzle
    // as the 128-wide kernels (slot s of bank row R stored at s ^ (R & 15), applied to the SOURCE address)
    unsigned a_vo[4], w_vo[4], g_chk[4];
    int g_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = ((i * 2 + wave) << 6) + lane;
        const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
        int am = m0 + 2 * R + (sx >> 3);
        am = am < p.M ? am : p.M - 1;
        g_row[i] = am;
        g_chk[i] = (sx & 7) * 16;
        a_vo[i] = GATHER ? 0x80000000u : ((unsigned)am * (unsigned)p.lda + (sx & 7) * 8) * 2;
        w_vo[i] = ((unsigned)(n0 + 2 * R + (sx >> 3)) * (unsigned)p.ldw + (sx & 7) * 8) * 2;
    }
    const int tps = GATHER ? p.seg_k / GEMM_BK : 1;               // K-tiles per gather segment
    const int frow = lane & 31, fchk = lane >> 5;
    unsigned a_rd[4], b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_rd[ks] = gemm_lds_off(wave * 32 + frow, ks * 2 + fchk);
        b_rd[ks] = 8192 + gemm_lds_off(frow, ks * 2 + fchk);       // second 32 W rows: + 4096 B
    }
    auto stage = [&](int kt) {
        const unsigned lds_buf = (unsigned)(kt & (GEMMS_STAGES - 1)) * GEMMS_STAGE_BYTES;
        unsigned ka = (unsigned)kt * (GEMM_BK * 2);
        const unsigned kw = ka;
        if (GATHER) {
            const int seg = kt / tps, kl = kt - seg * tps;
            if (kl == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = p.a_idx[(size_t)seg * p.idx_ld + g_row[i]];
                    a_vo[i] = r < 0 ? 0x80000000u : (unsigned)r * (unsigned)p.lda * 2u + g_chk[i];
                }
            }
            ka = (unsigned)kl * (GEMM_BK * 2);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + ((i * 2 + wave) << 10)),
                                                     16, a_vo[i], ka, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + 8192 + ((i * 2 + wave) << 10)),
                                                     16, w_vo[i], kw, 0, 0);
    };
    // prologue: three K-tiles in flight
#pragma unroll
    for (int s = 0; s < GEMMS_STAGES - 1; ++s)
        if (s < nt) stage(s);
    for (int kt = 0; kt < nt; ++kt) {
        // this wave's pieces of tile kt have landed once at most (tiles issued after kt) x 8 loads are outstanding
        const int newer = nt - 1 - kt < GEMMS_STAGES - 2 ? nt - 1 - kt : GEMMS_STAGES - 2;
        if (newer >= 2) VL2_WAIT_VMCNT(16); else if (newer == 1) VL2_WAIT_VMCNT(8); else VL2_WAIT_VMCNT(0);
        VL2_PHASE_BARRIER();                                        // everyone's pieces landed; buffer (kt-1)&3 is free
        if (kt + GEMMS_STAGES - 1 < nt) stage(kt + GEMMS_STAGES - 1);
        const unsigned lds_buf = (unsigned)(kt & (GEMMS_STAGES - 1)) * GEMMS_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 af = *(const bf16x8*)(vl2_smem + lds_buf + a_rd[ks]);
            const bf16x8 b0 = *(const bf16x8*)(vl2_smem + lds_buf + b_rd[ks]);
            const bf16x8 b1 = *(const bf16x8*)(vl2_smem + lds_buf + b_rd[ks] + 4096);
            acc[0] = VL2_MFMA32(af, b0, acc[0]);
            acc[1] = VL2_MFMA32(af, b1, acc[1]);
        }
    }
    VL2_WAIT_LGKMCNT0();
    VL2_PHASE_BARRIER();

    // ---- epilogue: one 32 x 64 fp32 patch per wave -> rows (same fused bias / activation / residual as the 128-wide kernel)
    float* ep = (float*)vl2_smem + wave * (32 * 68);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ep[row * 68 + ni * 32 + (lane & 31)] = acc[ni][r];
        }
    float* rowtab = (float*)vl2_smem + 2 * (32 * 68);
    gemm_park_row_stats(p, rowtab, rst, tid, GEMMS_BM);
    __syncthreads();
    gemm_store_patch<ACT, false, false, false>(p, ep, m0 + wave * 32, n0, lane, rowtab, wave * 32);
}


// ------------------------------------------------------------------------------------------------------------------
// gemm_l8 "one-round" form: the 128 x 128 tile for grids of at most one tile per CU.  There the 2-stage 4-wave kernel runs one
// wave per SIMD as a dependent chain (DMA -> barrier -> ds_read -> MFMA: ~0.64 us per K-tile however idle the chip is) and a
// second workgroup per CU never arrives to hide it.  Here the tile is split over EIGHT waves (wave tile 32 x 64, two waves per
// SIMD: one's fragment reads overlap the other's MFMAs) and, with one workgroup per CU, the LDS holds a 4-stage 128 KiB ring
// (three K-tiles of LDS-DMA in flight, counted vmcnt, ONE raw barrier per K-tile).  Measured (scripts/kernel_bench.py
// --frames 8): 845x4096x4096 43.7 -> 35.0 us, 945x4096x14336 152.6 -> 127.0 us; a 4-wave deep-ring variant only reached
// 39.5 / 135.6.  Same K order as every other kernel -> same bits.
#define GEMML_STAGES 4
#define GEMML_STAGE_BYTES 32768
#define GEMML_LDS_BYTES (GEMML_STAGES * GEMML_STAGE_BYTES)

// bid / nwg: this workgroup's index and the workgroup count of the tile set it belongs to (the whole grid for gemm_l8_bf16_kernel;
// the small-tile part of gemm_mix_bf16_kernel)
template <int ACT, bool SWIGLU, bool OUT_F32>
__device__ __forceinline__ void gemm_l8_body(const GemmArgs& p, int bid, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                    // wm 0..3 (32 rows each), wn 0..1 (64 columns each)
    const int t = xcd_remap(bid, nwg);
    const int tm = t % p.tiles_m, tn = t / p.tiles_m;            // consecutive workgroups share a W panel
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int nt = p.K / GEMM_BK;
    const f32x2 rst = gemm_row_stats(p, m0, tid, GEMM_BM);

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    unsigned a_vo[2], w_vo;                                     // 2 A pieces + 2 W pieces per wave (piece i = +64 rows)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = ((i * 8 + wave) << 6) + lane;
        const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
        int am = m0 + 2 * R + (sx >> 3);
        am = am < p.M ? am : p.M - 1;
        a_vo[i] = ((unsigned)am * (unsigned)p.lda + (sx & 7) * 8) * 2;
        if (i == 0) w_vo = ((unsigned)(n0 + 2 * R + (sx >> 3)) * (unsigned)p.ldw + (sx & 7) * 8) * 2;
    }
    const unsigned w_step = 128u * (unsigned)p.ldw;
    const int frow = lane & 31, fchk = lane >> 5;
    unsigned a_rd[4], b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_rd[ks] = gemm_lds_off(wm * 32 + frow, ks * 2 + fchk);
        b_rd[ks] = 16384 + gemm_lds_off(wn * 64 + frow, ks * 2 + fchk);
    }
    auto stage = [&](int kt) {
        const unsigned lds_buf = (unsigned)(kt & (GEMML_STAGES - 1)) * GEMML_STAGE_BYTES, kb = (unsigned)kt * (GEMM_BK * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + ((i * 8 + wave) << 10)),
                                                     16, a_vo[i], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + lds_buf + 16384 + ((i * 8 + wave) << 10)),
                                                     16, w_vo, kb + i * w_step, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < GEMML_STAGES - 1; ++s)
        if (s < nt) stage(s);
    for (int kt = 0; kt < nt; ++kt) {
        const int newer = nt - 1 - kt < GEMML_STAGES - 2 ? nt - 1 - kt : GEMML_STAGES - 2;
        if (newer >= 2) VL2_WAIT_VMCNT(8); else if (newer == 1) VL2_WAIT_VMCNT(4); else VL2_WAIT_VMCNT(0);
        VL2_PHASE_BARRIER();                                        // everyone's pieces of tile kt landed; buffer (kt-1)&3 is free
        if (kt + GEMML_STAGES - 1 < nt) stage(kt + GEMML_STAGES - 1);
        const unsigned lds_buf = (unsigned)(kt & (GEMML_STAGES - 1)) * GEMML_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 af = *(const bf16x8*)(vl2_smem + lds_buf + a_rd[ks]);
            const bf16x8 b0 = *(const bf16x8*)(vl2_smem + lds_buf + b_rd[ks]);
            const bf16x8 b1 = *(const bf16x8*)(vl2_smem + lds_buf + b_rd[ks] + 4096);
            acc[0] = VL2_MFMA32(af, b0, acc[0]);
            acc[1] = VL2_MFMA32(af, b1, acc[1]);
        }
    }
    VL2_WAIT_LGKMCNT0();
    VL2_PHASE_BARRIER();

    float* ep = (float*)vl2_smem + wave * (32 * 68);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ep[row * 68 + ni * 32 + (lane & 31)] = acc[ni][r];
        }
    float* rowtab = (float*)vl2_smem + 8 * (32 * 68);
    gemm_park_row_stats(p, rowtab, rst, tid, GEMM_BM);
    __syncthreads();
    gemm_store_patch<ACT, SWIGLU, OUT_F32>(p, ep, m0 + wm * 32, n0 + wn * 64, lane, rowtab, wm * 32);
    if constexpr (!SWIGLU && !OUT_F32) gemm_rows_ticket<512>(p, tm, m0, GEMM_BM, tid);
}
template <int ACT, bool SWIGLU, bool OUT_F32>
__global__ __launch_bounds__(512, 1) void gemm_l8_bf16_kernel(GemmArgs p) {
    gemm_l8_body<ACT, SWIGLU, OUT_F32>(p, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------
// gemm_mix: ONE launch for a row-split GEMM (vl2_abi.hip m_split_rows): workgroups [0, n_big) run the 256 x 256 ping-pong body on the
// leading rows (`pb`), workgroups [n_big, grid) the one-round 128 x 128 body on the tail rows (`ps`).  Workgroups are dispatched in
// index order, so the small tiles start on the CUs that run out of big tiles during the last, partially filled round of big tiles
// (gate/up at S = 1621: 672 big tiles = 2.625 rounds, 224 tail tiles) instead of in a second launch behind it.  Same bodies -> same bits.
template <int ACT, bool SWIGLU, bool TR, bool WEAVE4 = false>
__global__ __launch_bounds__(512, 2) void gemm_mix_bf16_kernel(GemmArgs pb, GemmArgs ps, int n_big) {
    if ((int)blockIdx.x < n_big) gemm4_body<ACT, SWIGLU, false, TR, -1, 256, false, WEAVE4>(pb, blockIdx.x, n_big);
    else gemm_l8_body<ACT, SWIGLU, false>(ps, (int)blockIdx.x - n_big, (int)gridDim.x - n_big);
}

// gemm7 "fill-the-round": a (128 + 32 R1) x 128 block tile (R1 = 3: 224 rows, R1 = 2: 192 rows) for GEMMs whose grid is ONE round of
// workgroups on the 256 CUs whatever the tile shape -- the decoder's o / down projections at S = 1621 (HF:modeling_mistral.py
// MistralAttention.o_proj / MistralMLP.down_proj), the STC 1x1 convolutions, readout and Conv3d taps on the 1521 output positions
// (videollama2/model/projector.py:164-187,208-214).
//
// Why (profiles/r04_experiments.md section 8): a one-round grid takes one full tile time however badly it fills the
// chip.  M = 1621, N = 4096 is 51 x 128 blocks of 32 x 32 = 25.5 per CU; the 128 x 256 kernel (gemm3) covers it with 13 x 16 = 208 tiles
// of 32 blocks (81 % of the CUs busy, 32 blocks of makespan), the 224 x 128 tile with 8 x 32 = 256 tiles of 28 blocks; M = 1521 is
// 48 x 128 blocks = 24 per CU: 192 tiles of 32 blocks before, 8 x 32 = 256 tiles of 24 blocks (192 x 128) here.  Splitting K instead
// would change the summation association with the grid, i.e. give up "a row's bits do not depend on M" (sharded == unsharded, batched ==
// one by one): every kernel of this library accumulates a tile's K range in ONE chain of 16-deep MFMA steps, and so does this one.
//
// Shape: 8 waves = two ping-pong groups as in gemm3 (waves w and w + 4 share a SIMD; one group issues only MFMAs while the other issues
// only LDS reads and LDS-DMA, raw s_barrier between the phases), but the groups own DIFFERENT row ranges so that every SIMD carries
// 4 + R1 accumulator blocks (7 is prime: no rectangular 8-way split exists):
//   group 0: rows [0, 128) x 128 columns, its four waves 2 x 2 (wave tile 64 x 64 = 2 x 2 blocks, 4 fragment reads per 4 MFMAs),
//   group 1: rows [128, 128 + 32 R1) x 128 columns, wave w owns the 32-column strip w (R1 x 1 blocks, R1 + 1 reads per R1 MFMAs).
// K-tile 64, 3-stage LDS ring of (A (128 + 32 R1) x 128 B | W 128 x 128 B) = 44 / 40 KiB per stage, same bank swizzle as the 128-wide
// kernels (gemm_lds_off).  A stage is 44 / 40 LDS-DMA pieces of 1 KiB; wave slot s issues pieces s, s + 8, ... -- group 1 (the shorter MFMA
// phase, so the longer load phase of its partner... is group 0's) takes the low slots, i.e. the sixth piece where there is one.
// Epilogue: the accumulators of all waves go to ONE fp32 image [BM][132] in LDS (the ring is dead by then), then every wave stores
// 8-row x 64-column passes through the same arithmetic as gemm_store_patch (bias / norm / activation / residual / statistics in that
// order) -> the same bits as every other kernel; the image decouples the store mapping from the odd MFMA mapping, so residual rows stay
// 16-B row-contiguous loads and the statistics keep their 64-column octet association.
#pragma once
#include "k_gemm.h"

#define GEMM7_BN 128
#define GEMM7_IMG_LD 132                                      // fp32 row stride of the epilogue image (128 + 4: lanes 32-63 land 16 banks off)
template <int R1> struct Gemm7Geo {
    static constexpr int BM = 128 + 32 * R1;
    static constexpr int NA = BM / 8;                          // A pieces (8 tile rows of 128 B each) per stage
    static constexpr int NPIECE = NA + 16;                     // + 16 W pieces
    static constexpr int STAGE = NPIECE * 1024;
    static constexpr int RING = 3 * STAGE;
    static constexpr int IMG = BM * GEMM7_IMG_LD * 4 + BM * 8; // epilogue image + row table
    static constexpr int LDS_BYTES = RING > IMG ? RING : IMG;
    static constexpr int NPW = (NPIECE + 7) / 8;               // most pieces a wave issues per stage
};

// 8 rows x 64 columns per pass: wave w stores column half (w & 1), row groups (w >> 1) + 4 j.  Arithmetic per element = gemm_store_patch's
// (non-SwiGLU, non-remap), in the same order: ((acc [norm]) + bias) -> activation -> + residual -> bf16; statistics from the stored bf16
// values, 8 columns per lane, octet_sum over the row's 8 lanes.
template <int ACT, bool OUT_F32, int NPASS>
__device__ __forceinline__ void gemm7_store_rows(const GemmArgs& p, const float* img, const float* rowtab, int m0, int n0, int wave, int lane) {
#pragma clang fp reassociate(off)
    const int cg = (lane & 7) * 8;
    const int ch = wave & 1, rg0 = wave >> 1;
    const int n = n0 + ch * 64 + cg;
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0, cs0 = bias0, cs1 = bias0;
    if (p.bias) { bias0 = *(const f32x4*)(p.bias + n); bias1 = *(const f32x4*)(p.bias + n + 4); }
    if (p.norm == 2) { cs0 = *(const f32x4*)(p.w_colsum + n); cs1 = *(const f32x4*)(p.w_colsum + n + 4); }
    u32x4 rv_[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {                     // the residual rows of every pass are requested up front (rows past M clamp)
        const int m = m0 + 8 * (rg0 + 4 * pass) + (lane >> 3);
        const int mc = m < p.M ? m : p.M - 1;
        if (p.res) rv_[pass] = *(const u32x4*)(p.res + (size_t)mc * p.ldres + n);
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int row = 8 * (rg0 + 4 * pass) + (lane >> 3);
        const int m = m0 + row;
        const bool live = m < p.M;
        float st_s = 0.f, st_q = 0.f;
        if (live) {
            float v[8];
            float mu = 0.f, rs = 1.f;
            if (p.norm) { mu = rowtab[2 * row]; rs = rowtab[2 * row + 1]; }
            const f32x4 x0 = *(const f32x4*)(img + row * GEMM7_IMG_LD + ch * 64 + cg), x1 = *(const f32x4*)(img + row * GEMM7_IMG_LD + ch * 64 + cg + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = x0[j]; v[4 + j] = x1[j]; }
            if (p.norm == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = __builtin_fmaf(-mu, cs0[j], v[j]) * rs; v[4 + j] = __builtin_fmaf(-mu, cs1[j], v[4 + j]) * rs; }
            } else if (p.norm == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= rs;
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
            if (p.res) {
                float rf[8];
                unpack8(rv_[pass], rf);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rf[j];
            }
            if (OUT_F32) {
                float* c = (float*)p.C + (size_t)m * p.ldc + n;
                f32x4 o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o0[j] = v[j]; o1[j] = v[4 + j]; }
                *(f32x4*)c = o0;
                *(f32x4*)(c + 4) = o1;
            } else {
                const u32x4 packed = pack8(v);
                *(u32x4*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = packed;
                if (p.stats_out) {                                 // statistics of the row AS STORED
                    float rf[8];
                    unpack8(packed, rf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { st_s += rf[j]; st_q = __builtin_fmaf(rf[j], rf[j], st_q); }
                }
            }
        }
        if (!OUT_F32 && p.stats_out) {                             // wave-uniform: the 8 lanes of a row fold their partials
            st_s = octet_sum(st_s);
            st_q = octet_sum(st_q);
            if (live && (lane & 7) == 0) {
                gemm_stat_put(p, p.stats_out + ((size_t)m * p.stats_out_np + ((n0 + ch * 64) >> 6)) * 2, st_s, st_q);
            }
        }
    }
}

// one group's main loop.  G0: 2 x 2 blocks (frag[ks][0..1] = A, [2..3] = W); group 1: R1 x 1 blocks (frag[ks][0..R1-1] = A, [3] = W).
// WEAVE: the LDS-DMA pieces of tile t + 2 are issued from the wave's own MFMA(t) phase, one piece behind every second MFMA (an in-order
// wave that streams MFMAs has idle issue slots: a piece costs ~60 cycles there against 100-185 in a load phase that also carries 16
// ds_read_b128 -- MI355X_MICROARCH.md "LDS-DMA piece issue cost"), and the load phase keeps the fragment reads only.  The pieces then have
// one phase of flight (retired by vmcnt(0) at the end of the wave's next load phase) instead of two.
// The same loop on v_mfma_f32_16x16x32_bf16 (round 6, the o / down projections of the decoder prefill: VL2_GEMM_MFMA16): a K-tile is two k-steps of the
// instruction; group 0's wave tile 64 x 64 = 4 x 4 blocks of 16 x 16 (4 + 4 fragment reads per 16 MFMAs), group 1's strip 32 R1 x 32 = 2 R1 x 2 blocks
// (2 R1 + 2 reads per 4 R1 MFMAs) -- the same number of fragment reads per K-tile as the 32 x 32 x 16 form, twice the MFMA count.  Fragment (block of 16
// rows) = rows 16 b + (lane & 15), 16-B chunk 4 kk + (lane >> 4): gemm9_body's k index function, k ascending, so a dot product is the same sequence of
// 32-product accumulation steps in every kernel of the 16 x 16 x 32 set (a row's bits do not depend on which of them computes it).
// a_rd / b_rd: [2 kk + parity] = LDS offset of block `parity`'s fragment at k-step kk; block b is + (b >> 1) * 4096 B from the entry of its parity (the
// image's swizzle repeats every 32 rows: gemm_lds_off xors the slot with (row >> 1) & 15).
template <int R1, bool G0, class Dma>
__device__ __forceinline__ void gemm7_loop16(unsigned char* smem, f32x4 (&acc)[16], const unsigned (&a_rd)[4], const unsigned (&b_rd)[4], int nt, int npw,
                                             Dma&& issue_dma) {
    constexpr int STAGE = Gemm7Geo<R1>::STAGE, NPW = Gemm7Geo<R1>::NPW;
    constexpr int NA_ = G0 ? 4 : 2 * R1, NB_ = G0 ? 4 : 2;          // 16-row blocks of A / W in the wave tile
    bf16x8 fa[2][NA_], fb[2][NB_];
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 2 < nt;
        if (more) issue_dma(t + 2, 0, NPW);
        const unsigned st = (unsigned)(t % 3) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < NA_; ++i) fa[kk][i] = *(const bf16x8*)(smem + a_rd[2 * kk + (i & 1)] + st + (i >> 1) * 4096);
#pragma unroll
            for (int j = 0; j < NB_; ++j) fb[kk][j] = *(const bf16x8*)(smem + b_rd[2 * kk + (j & 1)] + st + (j >> 1) * 4096);
        }
        if (more) { if (npw == 6) VL2_WAIT_VMCNT(6); else VL2_WAIT_VMCNT(5); }
        else VL2_WAIT_VMCNT(0);
        VL2_WAIT_LGKMCNT0();
        VL2_PHASE_BARRIER();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < NA_; ++i)
#pragma unroll
                for (int j = 0; j < NB_; ++j) acc[i * NB_ + j] = VL2_MFMA16(fa[kk][i], fb[kk][j], acc[i * NB_ + j]);
        VL2_PHASE_BARRIER();
    }
}

template <int R1, bool G0, bool WEAVE, class Dma>
__device__ __forceinline__ void gemm7_loop(unsigned char* smem, f32x16 (&acc)[4], const unsigned (&a_rd)[4], const unsigned (&b_rd)[4], int nt, int npw,
                                           Dma&& issue_dma) {
    constexpr int STAGE = Gemm7Geo<R1>::STAGE, NPW = Gemm7Geo<R1>::NPW;
    constexpr int NM = G0 ? 16 : 4 * R1;                           // MFMAs of a phase
    constexpr int WSTEP = NM >= 2 * NPW ? 2 : 1;                   // a piece behind every WSTEP-th MFMA
    bf16x8 frag[4][4];
    for (int t = 0; t < nt; ++t) {
        // ---------------- LOAD(t): memory work only
        const bool more = t + 2 < nt;
        if constexpr (!WEAVE) { if (more) issue_dma(t + 2, 0, NPW); }
        const unsigned st = (unsigned)(t % 3) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned ab = a_rd[ks] + st, bb = b_rd[ks] + st;
            if constexpr (G0) {
                frag[ks][0] = *(const bf16x8*)(smem + ab);
                frag[ks][1] = *(const bf16x8*)(smem + ab + 4096);
                frag[ks][2] = *(const bf16x8*)(smem + bb);
                frag[ks][3] = *(const bf16x8*)(smem + bb + 4096);
            } else {
#pragma unroll
                for (int i = 0; i < R1; ++i) frag[ks][i] = *(const bf16x8*)(smem + ab + i * 4096);
                frag[ks][3] = *(const bf16x8*)(smem + bb);
            }
        }
        // tile t + 1 landed (this wave's pieces of it); without WEAVE the pieces just issued stay in flight
        if (!WEAVE && more) { if (npw == 6) VL2_WAIT_VMCNT(6); else VL2_WAIT_VMCNT(5); }
        else VL2_WAIT_VMCNT(0);
        VL2_WAIT_LGKMCNT0();
        VL2_PHASE_BARRIER();
        // ---------------- MFMA(t): matrix work (+ the woven LDS-DMA issue)
#pragma unroll
        for (int idx = 0; idx < NM; ++idx) {
            if constexpr (G0) {
                const int ks = idx >> 2, i = (idx >> 1) & 1, j = idx & 1;
                acc[i * 2 + j] = VL2_MFMA32(frag[ks][i], frag[ks][2 + j], acc[i * 2 + j]);
            } else {
                const int ks = idx / R1, i = idx % R1;
                acc[i] = VL2_MFMA32(frag[ks][i], frag[ks][3], acc[i]);
            }
            if constexpr (WEAVE) {
                if (idx % WSTEP == WSTEP - 1 && idx / WSTEP < NPW) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) issue_dma(t + 2, idx / WSTEP, idx / WSTEP + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        VL2_PHASE_BARRIER();
    }
}

// (a __device__ body: the host pass of hipcc drops a __global__ template whose own body holds device-only inline asm -- VL2_PIN2 -- without
//  a diagnostic, and the launch then fails to link)
template <int ACT, bool OUT_F32, bool GATHER, int R1, bool WEAVE = false, bool M16 = false>
__device__ __forceinline__ void gemm7_body(const GemmArgs& p) {
    static_assert(R1 == 2 || R1 == 3, "gemm7: 192- or 224-row tiles");
    static_assert(!(M16 && (WEAVE || GATHER)), "gemm7: the 16 x 16 x 32 form is built plain (no gather, load-phase issue)");
    using Geo = Gemm7Geo<R1>;
    constexpr int BM = Geo::BM, NA = Geo::NA, NPIECE = Geo::NPIECE, STAGE = Geo::STAGE, NPW = Geo::NPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;                      // waves w and w + 4 share a SIMD

    // XCD-aware order: every XCD walks a contiguous run of tiles, 4 tile-rows deep (32 tiles = 4 row tiles x 8 W panels per L2)
    const int t0 = xcd_remap(blockIdx.x, gridDim.x);
    const int grp_sz = 4 * p.tiles_n;
    const int first_m = (t0 / grp_sz) * 4;
    const int gm = (p.tiles_m - first_m) < 4 ? (p.tiles_m - first_m) : 4;
    const int tm = first_m + (t0 % grp_sz) % gm, tn = (t0 % grp_sz) / gm;
    const int m0 = tm * BM, n0 = tn * GEMM7_BN;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    // this wave's LDS-DMA pieces of a stage: q = wslot + 8 i < NPIECE; q < NA: A rows [8 q, +8), else W rows [8 (q - NA), +8).  Group 1 takes
    // the low slots (six pieces of 44): its partner's MFMA phase, which covers this load phase, is the longer one.
    const int wslot = (wave + 4) & 7;
    const int npw = (NPIECE - wslot + 7) / 8;
    unsigned vo[NPW], g_chk[NPW];
    int g_row[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int q = wslot + 8 * i;
        const bool is_a = q < NA;
        const int slot = ((is_a ? q : q - NA) << 6) + lane;
        const int R = slot >> 4, sx = (slot & 15) ^ (R & 15);
        const int row = 2 * R + (sx >> 3);
        g_chk[i] = (sx & 7) * 16;
        g_row[i] = 0;
        if (is_a) {
            int am = m0 + row;
            am = am < p.M ? am : p.M - 1;
            g_row[i] = am;
            vo[i] = GATHER ? 0x80000000u : (unsigned)am * (unsigned)p.lda * 2u + g_chk[i];
        } else {
            vo[i] = (unsigned)(n0 + row) * (unsigned)p.ldw * 2u + g_chk[i];
        }
    }
    const int tps = GATHER ? p.seg_k / GEMM_BK : 1;                // K-tiles per gather segment
    auto issue_dma = [&](int kt, int i0, int i1) {              // pieces [i0, i1) of this wave's share of K-tile kt
        const unsigned st = (unsigned)(kt % 3) * STAGE, kw = (unsigned)kt * (GEMM_BK * 2);
        unsigned ka = kw;
        if constexpr (GATHER) {
            // the A row of (K segment, m) comes from the index table; a missing tap (index < 0) is an out-of-range buffer offset = zeros
            const int seg = kt / tps, kl = kt - seg * tps;
            if ((kl == 0 || kt == 0) && i0 == 0) {
#pragma unroll
                for (int i = 0; i < NPW; ++i)
                    if (wslot + 8 * i < NA) {
                        const int r = p.a_idx[(size_t)seg * p.idx_ld + g_row[i]];
                        vo[i] = r < 0 ? 0x80000000u : (unsigned)r * (unsigned)p.lda * 2u + g_chk[i];
                    }
            }
            ka = (unsigned)kl * (GEMM_BK * 2);
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int q = wslot + 8 * i;
            if (i >= i0 && i < i1 && q < NPIECE) {
                if (q < NA)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + (q << 10)), 16, vo[i], ka, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + (q << 10)), 16, vo[i], kw, 0, 0);
            }
        }
    };

    f32x16 acc[4];
    f32x4 acc16[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / GEMM_BK;
    const int frow = M16 ? (lane & 15) : (lane & 31), fchk = M16 ? (lane >> 4) : (lane >> 5);
    unsigned a_rd[4], b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                                 // (16 x 16 x 32: entry 2 kk + parity = k-step kk, 16-row block `parity` of the wave tile)
        const int chunk = M16 ? (ks >> 1) * 4 + fchk : ks * 2 + fchk;
        const int r16 = M16 ? (ks & 1) * 16 : 0;
        a_rd[ks] = gemm_lds_off((grp == 0 ? (w4 >> 1) * 64 : 128) + r16 + frow, chunk);
        b_rd[ks] = BM * 128 + gemm_lds_off((grp == 0 ? (w4 & 1) * 64 : w4 * 32) + r16 + frow, chunk);
    }
    issue_dma(0, 0, NPW);
    if (nt > 1) issue_dma(1, 0, NPW);
    // the rows' (mean, rstd) (norm-carrying GEMMs) ride behind the ring fill, as in gemm4
    f32x2 rst = gemm_row_stats(p, m0, tid, BM);
    VL2_PIN2(rst[0], rst[1]);
    if (nt > 1) { if (npw == 6) VL2_WAIT_VMCNT(6); else VL2_WAIT_VMCNT(5); }
    else VL2_WAIT_VMCNT(0);
    VL2_PHASE_BARRIER();

    if constexpr (M16) {
        if (grp == 0) {
            gemm7_loop16<R1, true>(vl2_smem, acc16, a_rd, b_rd, nt, npw, issue_dma);
            VL2_PHASE_BARRIER();
        } else {
            VL2_PHASE_BARRIER();
            gemm7_loop16<R1, false>(vl2_smem, acc16, a_rd, b_rd, nt, npw, issue_dma);
        }
    } else if (grp == 0) {
        gemm7_loop<R1, true, WEAVE>(vl2_smem, acc, a_rd, b_rd, nt, npw, issue_dma);
        VL2_PHASE_BARRIER();
    } else {
        VL2_PHASE_BARRIER();
        gemm7_loop<R1, false, WEAVE>(vl2_smem, acc, a_rd, b_rd, nt, npw, issue_dma);
    }

    // ---- epilogue: every wave's blocks -> the fp32 image (the ring is dead: all operands were in registers before the last barrier)
    float* img = (float*)vl2_smem;
    float* rowtab = img + BM * GEMM7_IMG_LD;
    if constexpr (M16) {                        // register r of block (i, j) = C[16 i + 4 (lane >> 4) + r][16 j + (lane & 15)]
        const int er = 4 * (lane >> 4), ec = lane & 15;
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        img[((w4 >> 1) * 64 + i * 16 + er + r) * GEMM7_IMG_LD + (w4 & 1) * 64 + j * 16 + ec] = acc16[i * 4 + j][r];
        } else {
#pragma unroll
            for (int i = 0; i < 2 * R1; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        img[(128 + i * 16 + er + r) * GEMM7_IMG_LD + w4 * 32 + j * 16 + ec] = acc16[i * 2 + j][r];
        }
    } else {
        const int er = 4 * (lane >> 5), ec = lane & 31;
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        img[((w4 >> 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + er) * GEMM7_IMG_LD + (w4 & 1) * 64 + j * 32 + ec] = acc[i * 2 + j][r];
        } else {
#pragma unroll
            for (int i = 0; i < R1; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    img[(128 + i * 32 + (r & 3) + 8 * (r >> 2) + er) * GEMM7_IMG_LD + w4 * 32 + ec] = acc[i][r];
        }
    }
    gemm_park_row_stats(p, rowtab, rst, tid, BM);
    __syncthreads();
    gemm7_store_rows<ACT, OUT_F32, BM / 32>(p, img, rowtab, m0, n0, wave, lane);
    if constexpr (!OUT_F32) gemm_rows_ticket<512>(p, tm, m0, BM, tid);
}
template <int ACT, bool OUT_F32, bool GATHER, int R1, bool WEAVE = false>
__global__ __launch_bounds__(512, 2) void gemm7_bf16_kernel(GemmArgs p) {
    gemm7_body<ACT, OUT_F32, GATHER, R1, WEAVE>(p);
}
// the same tile on v_mfma_f32_16x16x32_bf16 (bf16 output, no activation: the decoder's o / down projections under VL2_GEMM_MFMA16)
template <int R1>
__global__ __launch_bounds__(512, 2) void gemm7_16_bf16_kernel(GemmArgs p) {
    gemm7_body<ACT_NONE, false, false, R1, false, true>(p);
}

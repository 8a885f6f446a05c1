// Fused attention forward, second structure: K and V tiles arrive by LDS-DMA into a two-stage ring, V is consumed through the
// hardware transpose read.  Same arithmetic and the same transposed formulation as k_attn.h (S^T = K . Q^T with lane-local
// softmax statistics, O^T += V^T . P^T with the score registers as the B operand; HF eager_attention_forward of
// HF:models/clip/modeling_clip.py and HF:models/mistral/modeling_mistral.py, softmax in fp32), different data movement:
//
//   k_attn.h                                              here
//   K/V: global -> 32 VGPRs -> LDS (ds_write_b128 +       `buffer_load ... lds` (LDS-DMA, 16 B per lane) straight into the image:
//        16 transposing ds_write_b32 per thread and       no staging registers, no ds_write, no packing VALU; the per-lane SOURCE
//        tile, ~170 VALU of address / packing work)       offsets are loop-invariant, the tile position sits in the buffer base
//   V^T image built by the writes, one ds_read_b128      V stays row-major (4-key x 16-d blocks of 128 B) and the A fragment is two
//        per PV fragment                                  `ds_read_b64_tr_b16` from ONE base VGPR + immediate offsets
//   one LDS buffer, two barriers per KV tile              two stages, ONE barrier per tile: tile t+1 is in flight during tile t
//
// Why (profiles/r01_gemm_experiments.md, "Attention ablations" + r01_attn_pmc_counters.csv): the r01 kernels keep the matrix
// pipe 17 % (causal D = 128) / 24 % (ViT D = 64) busy; K/V staging is worth 24 % of the time, more than half of it the
// global-load/VALU side, and the longest causal q block is a chain of 26 dependent tiles at ~4.6 k cycles each.
//
// LDS images (per stage):
//   K  [64 keys][D] bf16, 16-B chunk c of row r at the swizzled position of k_attn.h (`attn_k_off`): conflict-free
//      ds_read_b128 fragments.  LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE chunk (guide rule 21).
//   V  [64/4 key quads][D/16 column blocks][4 keys][16 d] bf16 (128-B blocks).  A 32-lane half of a transpose read covers two
//      adjacent blocks = 256 contiguous bytes = every bank once.  ds_read_b64_tr_b16 semantics (per 16-lane group): lane m
//      supplies the address of 4 consecutive bf16; lane i receives element (i & 3) of lanes 4e + (i >> 2), e = 0..3 -- with lane m
//      pointing at row m >> 2, columns 4 (m & 3).. of a [4][16] block, lane i ends up with column i of rows 0..3: the V^T
//      fragment (d = lane, 4 consecutive keys) the PV MFMA wants as its A operand.
#pragma once
#include "dev_common.h"
#include "k_attn.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef VL2_WAIT_VMCNT
#define VL2_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))
#endif
#define VL2_ATTN2_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// LDS transpose read (see the header comment).  A plain device function: inside the kernel TEMPLATE the builtin's address-space
// cast is a dependent expression the host pass of hipcc rejects silently (the kernel's host stub is then never emitted).
__device__ __forceinline__ s16x4 lds_read_tr16(const unsigned char* lds_byte_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_byte_ptr);
}

// One 1-KiB LDS-DMA piece: 16 B per lane from buffer `rs` at byte offset `voff` into lds_dst + 16 * lane.  Plain (non-template)
// device function on purpose, like lds_read_tr16: with type-dependent arguments the builtin is only checked when the kernel
// template is instantiated, and the HOST pass of hipcc fails that check silently and drops the kernel's stub.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}

// ONES: the softmax denominators come off the matrix pipe -- one more MFMA per P fragment with an all-ones A operand accumulates
// sum_k P[q][k] (of the bf16-rounded P the numerator uses) in every row of a 32 x 32 block; the 32 adds per lane and tile and the
// cross-half shuffle go away.  Lab variant 4 of vl2_attn_fwd (scripts/attn_bench2.py); see profiles/r03_experiments.md section 5b.
// PIPE: the tile is worked in two 32-key halves and the two halves' matrix and vector work are interleaved IN the wave (MFMA and VALU share
// the issue port, and four resident waves were measured not to overlap them: profiles/r03_experiments.md section 5b): the QK^T MFMAs of half 1
// run between the exp / sum / pack instructions of half 0, the PV MFMAs of half 0 between those of half 1; the running maximum is updated per
// half (a rescale found in half 1 is applied to O after PV of half 0 has been issued).  Lab variant 5 of vl2_attn_fwd.
#ifndef VL2_PERMLANE32_SWAP_2
#define VL2_PERMLANE32_SWAP_2(a, b) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b))
#define VL2_PIN3(a, b, c) asm volatile("" :: "v"(a), "v"(b), "v"(c))     // a use the optimiser cannot move: pins the producers before this point
#endif
template <int D, bool CAUSAL, bool ONES = false, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void attn2_fwd_kernel(AttnArgs p) {
    static_assert(D == 64 || D == 128, "attn2: head_dim 64 or 128");
    constexpr int NKS = D / 16;                // k-steps of the QK^T MFMA chain
    constexpr int NDB = D / 32;                // 32-row d blocks of O^T
    constexpr int K_BYTES = 64 * D * 2, STAGE = 2 * K_BYTES;
    constexpr int PPW = K_BYTES / 1024 / 4;    // 1-KiB LDS-DMA pieces per wave, per operand and tile (D = 128: 4, D = 64: 2)
    constexpr int QUAD = (D / 16) * 128;       // bytes of one key quad in the V image
    __shared__ __attribute__((aligned(16))) unsigned char lds_mem[2 * STAGE];
    unsigned char* const lds = lds_mem;        // the lambdas below capture this pointer, not the __shared__ array itself (casting
                                               // the array to an LDS address space inside a lambda of a kernel TEMPLATE makes the host
                                               // pass of hipcc drop the kernel's stub without a diagnostic)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    // grid mapping as in k_attn.h (causal: longest q blocks first, then ascending, so a CU pairs a long block with a short one)
    int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (CAUSAL) {
        const int G = p.heads * p.batch, nqb = (p.nq + 127) >> 7;
        const int g = (int)blockIdx.x % G, r = (int)blockIdx.x / G;
        int nfirst = (256 + G - 1) / G;
        nfirst = nfirst < nqb ? nfirst : nqb;
        qb = r < nfirst ? nqb - 1 - r : r - nfirst;
        h = g % p.heads;
        b = g / p.heads;
    }
    const int hk = h / p.group;
    const int q0 = qb * 128;
    const bf16_t* Q = p.q + b * p.q_bs + h * p.q_hs;
    const bf16_t* K = p.k + b * p.k_bs + hk * p.k_hs;
    const bf16_t* V = p.v + b * p.v_bs + hk * p.v_hs;

    // Q^T fragments: lane (q, hi) holds Q[q][16*ks + 8*hi .. +7]
    const int qrow = q0 + wave * 32 + l31;
    const int qrow_c = qrow < p.nq ? qrow : p.nq - 1;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(Q + (size_t)qrow_c * p.q_rs + ks * 16 + hi * 8);

    int kmax = p.nk;
    if (CAUSAL) { const int lim = q0 + 128 + p.causal_off; kmax = lim < kmax ? lim : kmax; }
    const int ntiles = (kmax + 63) >> 6;

    // ---- LDS-DMA source offsets of this lane (bytes inside a tile; loop-invariant).  Piece pc = i * 4 + wave fills LDS bytes
    //      [pc * 1024, +1024): lane L lands at slot pc * 64 + L (16-B slots).
    unsigned koff[PPW], voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int slot = ((i * 4 + wave) << 6) + lane;
        int row, chunk;
        if (D == 128) { row = slot >> 4; chunk = (slot & 15) ^ (row & 15); }
        else { const int R = slot >> 4, s = (slot & 15) ^ (R & 15); row = 2 * R + (s >> 3); chunk = s & 7; }
        koff[i] = (unsigned)(row * p.k_rs + chunk * 8) * 2u;
        const int blk = slot >> 3, w8 = slot & 7;
        const int key = (blk / (D / 16)) * 4 + (w8 >> 1), d0 = (blk % (D / 16)) * 16 + (w8 & 1) * 8;
        voff[i] = (unsigned)(key * p.v_rs + d0) * 2u;
    }
    const int k_bytes = ((p.nk - 1) * p.k_rs + D) * 2, v_bytes = ((p.nk - 1) * p.v_rs + D) * 2;   // valid bytes behind K / V
    // rows past nk lie outside NUM_RECORDS and arrive as zeros (their scores are masked, their P is 0).
    // (descriptors declared with their type, not `auto`: see lds_dma16)
    auto dma_tile = [&](int t, unsigned so) {
        const int kskip = t * 64 * p.k_rs * 2, vskip = t * 64 * p.v_rs * 2;
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)K + kskip), 0, k_bytes > kskip ? k_bytes - kskip : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)V + vskip), 0, v_bytes > vskip ? v_bytes - vskip : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            lds_dma16(rsK, lds + so + ((i * 4 + wave) << 10), koff[i]);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            lds_dma16(rsV, lds + so + K_BYTES + ((i * 4 + wave) << 10), voff[i]);
    };

    // ---- fragment read bases (loop-invariant): K row l31 (+32 per kh: an immediate), V transpose-read base of this lane
    unsigned kbase[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kbase[ks] = (unsigned)attn_k_off<D>(l31, ks * 2 + hi);
    constexpr int KH_STEP = D == 128 ? 32 * 256 : 16 * 256;       // 32 keys further in the K image
    const int g4 = lane >> 4, m16 = lane & 15;
    const unsigned vbase = (unsigned)(K_BYTES + (g4 & 1) * 128 + (g4 >> 1) * QUAD + (m16 >> 2) * 32 + (m16 & 3) * 8);

    f32x16 oT[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
    float m = -1e30f, l = 0.f;      // m: running max in the exp2 domain (first tile always rescales: mt - m is huge)
    f32x16 oL;                      // ONES: every register of the lane = l of its query row
#pragma unroll
    for (int r = 0; r < 16; ++r) oL[r] = 0.f;
    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    auto compute_tile = [&](int t, unsigned so) {
        const int kv0 = t * 64;
        // S^T = K . Q^T
        f32x16 sT[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(lds + so + kh * KH_STEP + kbase[ks]);
                sT[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero16 : sT[kh], 0, 0, 0);
            }
        }
        // online softmax, exp2 domain (k_attn.h): lane owns keys kv0 + 32kh + (r&3) + 8(r>>2) + 4hi of row qrow
        const int wq0 = q0 + wave * 32;
        const bool need_mask = (kv0 + 64 > p.nk) || (CAUSAL && (kv0 + 63 > wq0 + p.causal_off));
        const float c = p.scale_log2e;
        float mt = -3.0e38f;
        if (need_mask) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < p.nk && (!CAUSAL || key <= qrow + p.causal_off);
                    sT[kh][r] = ok ? sT[kh][r] : -1e30f;           // raw domain; c > 0
                }
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sT[kh][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32)) * c;
        constexpr float THR = 6.0f;                                // deferred rescale (guide T13), as k_attn.h
        if (!__all(mt - m <= THR)) {
            const float m_new = fmaxf(fmaxf(m, mt), -1e28f);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[i][r] *= alpha;
            if constexpr (ONES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) oL[r] *= alpha;
            }
        }
        float rs = 0.f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sT[kh][r], c, -m));
                sT[kh][r] = pv;
                if constexpr (!ONES) rs += pv;
            }
        if constexpr (!ONES) l += rs + __shfl_xor(rs, 32);

        // O^T += V^T . P^T: P fragment = the score registers; V^T fragment = two transpose reads (keys kb..kb+3, kb+8..kb+11)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                u32x4 pw;
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[j] = pack2bf(sT[kh][ks2 * 8 + 2 * j], sT[kh][ks2 * 8 + 2 * j + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
                if constexpr (ONES) oL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pf, oL, 0, 0, 0);
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const unsigned a = so + vbase + (8 * kh + 4 * ks2) * QUAD + db * 256;
                    const s16x4 v0 = lds_read_tr16(lds + a), v1 = lds_read_tr16(lds + a + 2 * QUAD);
                    const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oT[db], 0, 0, 0);
                }
            }
    };

    // ---- PIPE: the same tile in two halves, matrix and vector work interleaved by hand (sched_barrier fences pin the order)
    auto compute_tile_pipe = [&](int t, unsigned so) {
        const int kv0 = t * 64;
        const int wq0 = q0 + wave * 32;
        const bool need_mask = (kv0 + 64 > p.nk) || (CAUSAL && (kv0 + 63 > wq0 + p.causal_off));
        const float c = p.scale_log2e;
        constexpr float THR = 6.0f;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 sT[2];
        u32x4 pw[2][2];
        bf16x8 kf[NKS];
        // (mask +) max of half kh over the lane's 16 keys and the partner lane's (lanes l, l + 32 hold the two key sets of row qrow)
        auto half_max = [&](int kh) -> float {
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < p.nk && (!CAUSAL || key <= qrow + p.causal_off);
                    sT[kh][r] = ok ? sT[kh][r] : -1e30f;
                }
            }
            float mt = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sT[kh][r]);
            float a = mt, b = mt;
            VL2_PERMLANE32_SWAP_2(a, b);                  // a = lower half's value, b = upper half's, in every lane
            return fmaxf(a, b) * c;
        };
        // exp / sum / pack of elements [first, first + n) of half kh (n even: whole bf16 pairs)
        auto exp_chunk = [&](int kh, int first, int n) {
#pragma unroll
            for (int r = first; r < first + n; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sT[kh][r], c, -m));
                sT[kh][r] = pv;
                l += pv;
            }
#pragma unroll
            for (int r = first; r < first + n; r += 2) pw[kh][r >> 3][(r & 7) >> 1] = pack2bf(sT[kh][r], sT[kh][r + 1]);
        };
        // ---- S of half 0; the K fragments of half 1 are requested behind it
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[ks] = *(const bf16x8*)(lds + so + kbase[ks]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) sT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks == 0 ? zero16 : sT[0], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[ks] = *(const bf16x8*)(lds + so + KH_STEP + kbase[ks]);
        float mt = half_max(0);
        if (!__all(mt - m <= THR)) {
            const float m_new = fmaxf(fmaxf(m, mt), -1e28f);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[i][r] *= alpha;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- region 1: S of half 1 between the exp / sum / pack of half 0.  One scheduling region; the sched_group_barrier pipeline
        //      below tells the machine scheduler the issue order: MFMA, then its share of the vector work (VALU mask 0x2 + transcendental
        //      mask 0x400), NKS times.  (Fences in source order are not enough: the pure exp2 / fma calls are regrouped before scheduling.)
        constexpr int VPM1 = (16 * 3 + 8) / NKS, TPM1 = 16 / NKS;      // vector / transcendental instructions per MFMA: 16 x (fma, add) + 8 cvt; 16 exp
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) sT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks == 0 ? zero16 : sT[1], 0, 0, 0);
        exp_chunk(0, 0, 16);
        VL2_PIN3(pw[0][0], pw[0][1], l);                  // a use INSIDE the region: the optimiser otherwise sinks the pure exp / pack work to PV
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM1 - TPM1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, TPM1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragments of half 0 (all of them) requested before the statistics of half 1
        s16x4 v0[2 * NDB], v1[2 * NDB];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const unsigned a = so + vbase + (4 * ks2) * QUAD + db * 256;
                v0[ks2 * NDB + db] = lds_read_tr16(lds + a);
                v1[ks2 * NDB + db] = lds_read_tr16(lds + a + 2 * QUAD);
            }
        mt = half_max(1);
        float alpha1 = 1.0f;
        const bool resc = !__all(mt - m <= THR);
        if (resc) {
            const float m_new = fmaxf(fmaxf(m, mt), -1e28f);
            alpha1 = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            l *= alpha1;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- region 2: PV of half 0 between the exp / sum / pack of half 1 (same pipeline, 2 NDB MFMAs)
        constexpr int VPM2 = (16 * 3 + 8) / (2 * NDB), TPM2 = 16 / (2 * NDB);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int st = ks2 * NDB + db;
                const bf16x8 vf = {v0[st][0], v0[st][1], v0[st][2], v0[st][3], v1[st][0], v1[st][1], v1[st][2], v1[st][3]};
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pw[0][ks2]), oT[db], 0, 0, 0);
            }
        exp_chunk(1, 0, 16);
        VL2_PIN3(pw[1][0], pw[1][1], l);
#pragma unroll
        for (int st = 0; st < 2 * NDB; ++st) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM2 - TPM2, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, TPM2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragments of half 1
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const unsigned a = so + vbase + (8 + 4 * ks2) * QUAD + db * 256;
                v0[ks2 * NDB + db] = lds_read_tr16(lds + a);
                v1[ks2 * NDB + db] = lds_read_tr16(lds + a + 2 * QUAD);
            }
        if (resc) {                                       // the maximum moved in half 1: O (now holding half 0's PV as well) follows
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[i][r] *= alpha1;
        }
        // ---- PV of half 1
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int st = ks2 * NDB + db;
                const bf16x8 vf = {v0[st][0], v0[st][1], v0[st][2], v0[st][3], v1[st][0], v1[st][1], v1[st][2], v1[st][3]};
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pw[1][ks2]), oT[db], 0, 0, 0);
            }
    };

    // ---- main loop: tile t lives in stage t & 1.  Top of tile t: this wave's pieces of tile t have landed (its only
    //      outstanding VMEM), the barrier makes everyone's pieces visible AND proves every wave is done reading the other stage
    //      (tile t-1), which the DMA of tile t+1 may therefore overwrite while tile t is computed.  One barrier per tile.
    // a wave whose 32 query rows all lie past nq (the last query block of a 577-row ViT frame: rows 65..127 of it) still moves its
    // share of every K/V tile and meets every barrier, but skips the arithmetic: its issue slots go to the waves it shares a SIMD with
    const bool live = q0 + wave * 32 < p.nq;
    dma_tile(0, 0);
    for (int t = 0; t < ntiles; t += 2) {
        VL2_WAIT_VMCNT(0);
        VL2_ATTN2_BARRIER();
        if (t + 1 < ntiles) dma_tile(t + 1, STAGE);
        if (live) { if constexpr (PIPE) compute_tile_pipe(t, 0); else compute_tile(t, 0); }
        if (t + 1 >= ntiles) break;
        VL2_WAIT_VMCNT(0);
        VL2_ATTN2_BARRIER();
        if (t + 2 < ntiles) dma_tile(t + 2, 0);
        if (live) { if constexpr (PIPE) compute_tile_pipe(t + 1, STAGE); else compute_tile(t + 1, STAGE); }
    }

    if constexpr (PIPE) {                 // l holds this lane's keys only: add the partner lane's (same query row, the other key sets)
        float a = l, b = l;
        VL2_PERMLANE32_SWAP_2(a, b);
        l = a + b;
    }
    if (qrow < p.nq) {
        const float inv = 1.0f / (ONES ? oL[0] : l);        // (PIPE: l was completed across the two half-waves just above)
        bf16_t* O = p.o + b * p.o_bs + h * p.o_hs + (size_t)qrow * p.o_rs;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = pack2bf(oT[db][4 * g] * inv, oT[db][4 * g + 1] * inv);
                w[1] = pack2bf(oT[db][4 * g + 2] * inv, oT[db][4 * g + 3] * inv);
                *(u32x2*)(O + db * 32 + 8 * g + 4 * hi) = w;
            }
    }
}

// Two further structures were built on these images, measured on MI355X and removed again (numbers and reading:
// profiles/r02_experiments.md): (a) the whole K/V of a (frame, head) resident in LDS (2 x 80 KiB), 8 waves walking q blocks
// with a software-pipelined tile loop -- ViT T=16 48.6 us vs 43.0 us here; (b) the D = 128 loop software-pipelined inside the
// wave (QK^T of tile t+1 before the softmax of tile t, three-stage ring, one wave per SIMD) -- S=1621 54.9 us vs 38.1 us here.
// Both are bound by VALU issue (about 250 VALU instructions per wave and tile against 16 / 32 MFMAs), not by staging latency.

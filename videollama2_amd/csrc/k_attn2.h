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

// (Two lab variants of this kernel -- the softmax denominators on the matrix pipe, and the tile in two halves with matrix and vector work
// interleaved in the wave -- were measured not faster (profiles/r03_experiments.md section 5b); their source is frozen in
// scripts/ubench/k_attn2_lab.h, outside the product.)
#ifndef VL2_PERMLANE32_SWAP_2
#define VL2_PERMLANE32_SWAP_2(a, b) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b))
#define VL2_PIN3(a, b, c) asm volatile("" :: "v"(a), "v"(b), "v"(c))     // a use the optimiser cannot move: pins the producers before this point
#endif
// CLS = true (full attention of a sequence whose row 0 is a class token, nq == nk == 1 + 64 n: the CLIP tower's 577 = 1 + 576 tokens; same
// result as CLS = false): the class token is PEELED off the tiling.  Its KEY becomes the initial online-softmax state of every query --
// m = s_cls, l = 1, O = V[0] from one 64-wide dot product on the vector ALU -- so the key tiles cover rows 1 .. nk - 1 exactly (nine
// 64-key tiles instead of ten, the tenth holding ONE key) and never need a mask; its QUERY is the only live row of a wave that holds no
// patch row (the patch queries fill whole 32-row waves: 576 = 18 x 32; unpeeled, that wave ran all ten tiles for row 576 alone).
// HF:modeling_clip.py:272 computes the same softmax.
//
// NS = 2 (round 6; 512 threads): TWO KEY STREAMS per query block.  Waves 0-3 walk the even 64-key tiles, waves 4-7 the odd ones, of the
// SAME 128 query rows, each stream with its own two-stage ring (dynamic LDS: 4 stages) and its own online-softmax state; after the
// last tile the odd stream hands (m, l, O^T) over through LDS and the even stream merges the two states (the flash-decoding combine,
// in fp32, inside the workgroup) and stores.  Why: the kernel's time is the longest DEPENDENT chain of tiles of one wave (causal
// S = 1621: 26 tiles of the last query block at ~2.9 k cycles each, the co-resident workgroup long gone; ViT: "10 us fixed" of the
// affine law in profiles/r03_experiments.md 5b) and two waves of one SIMD do not slow each other much below their sum -- so the chain
// is cut in two and both halves run on the same SIMDs.  A wave also skips a tile that the causal mask hides from all of its rows
// (every P of it is 0: same bits as computing it).
// SKIP = false (lab, variant 5): the one-stream kernels as they were before round 6's scheduling changes (that skip; the tower's grid order), for A/B runs.
template <int D, bool CAUSAL, bool CLS = false, int NS = 1, bool SKIP = true>
__global__ __launch_bounds__(256 * NS, (NS == 2 && D == 64) ? 4 : 2) void attn2_fwd_kernel(AttnArgs p) {
    static_assert(D == 64 || D == 128, "attn2: head_dim 64 or 128");
    static_assert(!(CLS && CAUSAL), "the class-token peel is for full attention");
    static_assert(NS == 1 || NS == 2, "attn2: one or two key streams");
    constexpr int NKS = D / 16;                // k-steps of the QK^T MFMA chain
    constexpr int NDB = D / 32;                // 32-row d blocks of O^T
    constexpr int K_BYTES = 64 * D * 2, STAGE = 2 * K_BYTES;
    constexpr int PPW = K_BYTES / 1024 / 4;    // 1-KiB LDS-DMA pieces per wave, per operand and tile (D = 128: 4, D = 64: 2)
    constexpr int QUAD = (D / 16) * 128;       // bytes of one key quad in the V image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st = NS == 2 ? wave_all >> 2 : 0;             // key stream of this wave: global tiles st, st + NS, ...
    const int wave = NS == 2 ? wave_all & 3 : wave_all;     // its 32-row group of the query block / its share of the stream's DMA
    unsigned char* lds_all;
    if constexpr (NS == 1) {
        __shared__ __attribute__((aligned(16))) unsigned char lds_mem[2 * STAGE];
        lds_all = lds_mem;
    } else {
        extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];      // NS * 2 * STAGE bytes
        lds_all = vl2_smem;
    }
    unsigned char* const lds = lds_all + st * (2 * STAGE);   // the lambdas below capture this pointer, not the __shared__ array itself
                                               // (casting the array to an LDS address space inside a lambda of a kernel TEMPLATE makes
                                               // the host pass of hipcc drop the kernel's stub without a diagnostic)
    const int hi = lane >> 5, l31 = lane & 31;
    // grid mapping (causal): longest q blocks first; see below for the order of the rest
    int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    // CLS (the tower): the query block is the SLOWEST grid index -- the ragged last block of every (head, frame) (64 patch rows + the class query: two and a bit
    // live waves of four) is dispatched last, so the launch's partially filled last round of workgroups is made of its cheapest ones
    if constexpr (CLS && SKIP && NS == 1) { qb = blockIdx.z; h = blockIdx.x; b = blockIdx.y; }
    if (CAUSAL) {
        const int G = p.heads * p.batch, nqb = (p.nq + 127) >> 7;
        const int g = (int)blockIdx.x % G, r = (int)blockIdx.x / G;
        int nfirst = (256 + G - 1) / G;
        nfirst = nfirst < nqb ? nfirst : nqb;
        // more workgroups than the chip holds at once (2 per CU): plain longest-first -- the cheap blocks make up the last, partially filled round (round 6:
        // S = 2973 92.9 -> 82.5 us, 64 heads at S = 1621 68.9 -> 55.3); all resident: the first 256 longest, then ascending, so that a CU pairs a long block with a short one
        qb = (SKIP && nqb * G > (NS == 2 ? 256 : 512)) ? nqb - 1 - r : (r < nfirst ? nqb - 1 - r : r - nfirst);
        if (SKIP && NS == 1 && nqb * G <= 512 && nqb > nfirst) {
            // all resident, two per CU at most: the (nfirst - n2) longest blocks should keep their CU to themselves, so the n2 blocks that will get a
            // partner go first (the dispatcher fills every CU's first slot before any second one: workgroup 256 + i joins workgroup i).  Round 6, on top of the
            // hidden-tile skip: S = 1621 39.6 -> 37.8 us, S = 1792 40.5 -> 39.7, 28 heads at S = 1452 32.4 -> 31.5 (profiles/r06_attn_alone_ab.jsonl)
            const int n2 = nqb - nfirst, alone = nfirst - n2;
            qb = r < n2 ? nqb - 1 - alone - r : r < nfirst ? nqb - 1 - (r - n2) : r - nfirst;
        }
        h = g % p.heads;
        b = g / p.heads;
    }
    const int hk = h / p.group;
    // CLS: query blocks over the patch rows.  The class query rides in the first DEAD wave of the last block (576 = 4.5 blocks: waves 2, 3
    // of block 4 hold no patch row) -- a workgroup of its own per (head, frame) costs more than the tenth key tile it saves (measured:
    // +256 workgroups = +6 us against -3 us); when the last block has no dead wave the launcher adds a block (qb == nqb_p) for it.
    const int nqb_p = (p.nq - 1 + 127) >> 7;
    const int cls_rem = (p.nq - 1) & 127;                              // patch rows in the last block (0 = it is full)
    const bool cls_in_last = cls_rem != 0 && cls_rem <= 96;
    const int cls_wave = CLS ? (cls_in_last ? (cls_rem + 31) >> 5 : 0) : -1;     // first dead wave of the last block / wave 0 of the extra block
    const int cls_qb = CLS ? (cls_in_last ? nqb_p - 1 : nqb_p) : -1;
    const bool cls_w = CLS && qb == cls_qb && wave == cls_wave;        // this wave holds the class query (row 0 of it) and nothing else
    const int q0 = CLS ? 1 + qb * 128 : qb * 128;
    const bf16_t* Q = p.q + b * p.q_bs + h * p.q_hs;
    const bf16_t* K = p.k + b * p.k_bs + hk * p.k_hs;
    const bf16_t* V = p.v + b * p.v_bs + hk * p.v_hs;

    // Q^T fragments: lane (q, hi) holds Q[q][16*ks + 8*hi .. +7]
    // CLS, the class-query wave: only its row 0 is a query; its other rows are dead (stored nowhere)
    const int qrow = cls_w ? (l31 == 0 ? 0 : p.nq) : q0 + wave * 32 + l31;
    const int qrow_c = qrow < p.nq ? qrow : p.nq - 1;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(Q + (size_t)qrow_c * p.q_rs + ks * 16 + hi * 8);

    int kmax = CLS ? p.nk - 1 : p.nk;                       // CLS: keys of the tiles = rows 1 .. nk - 1
    if (CAUSAL) { const int lim = q0 + 128 + p.causal_off; kmax = lim < kmax ? lim : kmax; }
    const int ntiles = (kmax + 63) >> 6;
    constexpr int KROW0 = CLS ? 1 : 0;                      // first K / V row of tile 0

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
    auto dma_tile = [&](int ts, unsigned so) {
        const int t = NS * ts + st;                             // global tile of this stream's ts-th
        const int kskip = (KROW0 + t * 64) * p.k_rs * 2, vskip = (KROW0 + t * 64) * p.v_rs * 2;
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
    if (CLS && (NS == 1 || st == 0)) {
        // (two key streams: the even stream starts from the class key, the odd one from the empty state)
        // the class token's key (row 0) as the initial state: s = q . K[0] (this lane holds half of its query's dims, the partner lane the
        // other half), p = exp2(s c - m) = 1 with m = s c, O^T = 1 * V[0]
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const u32x4 kc = *(const u32x4*)(K + ks * 16 + hi * 8);
            const u32x4 qc = __builtin_bit_cast(u32x4, qf[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part = __builtin_fmaf(e_lo(qc[j]), e_lo(kc[j]), part);
                part = __builtin_fmaf(e_hi(qc[j]), e_hi(kc[j]), part);
            }
        }
        m = (part + __shfl_xor(part, 32)) * p.scale_log2e;
        l = 1.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x2 vc = *(const u32x2*)(V + db * 32 + 8 * g + 4 * hi);
                oT[db][4 * g] = e_lo(vc[0]);
                oT[db][4 * g + 1] = e_hi(vc[0]);
                oT[db][4 * g + 2] = e_lo(vc[1]);
                oT[db][4 * g + 3] = e_hi(vc[1]);
            }
    }

    auto compute_tile = [&](int ts, unsigned so) {
        const int kv0 = (NS * ts + st) * 64;
        if (CAUSAL && SKIP && kv0 > q0 + wave * 32 + 31 + p.causal_off) return;        // hidden from all 32 rows of this wave: every P is 0 (same bits as computing it)
        // S^T = K . Q^T
        f32x16 sT[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(lds + so + kh * KH_STEP + kbase[ks]);
                sT[kh] = VL2_MFMA32(kf, qf[ks], ks == 0 ? zero16 : sT[kh]);
            }
        }
        // online softmax, exp2 domain (k_attn.h): lane owns keys kv0 + 32kh + (r&3) + 8(r>>2) + 4hi of row qrow
        const int wq0 = q0 + wave * 32;
        const int nk_t = CLS ? p.nk - 1 : p.nk;                 // keys covered by the tiles
        const bool need_mask = (kv0 + 64 > nk_t) || (CAUSAL && (kv0 + 63 > wq0 + p.causal_off));
        const float c = p.scale_log2e;
        float mt = -3.0e38f;
        if (need_mask) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < nk_t && (!CAUSAL || key <= qrow + p.causal_off);
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
        }
        float rs = 0.f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sT[kh][r], c, -m));
                sT[kh][r] = pv;
                rs += pv;
            }
        l += rs + __shfl_xor(rs, 32);

        // O^T += V^T . P^T: P fragment = the score registers; V^T fragment = two transpose reads (keys kb..kb+3, kb+8..kb+11)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                u32x4 pw;
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[j] = pack2bf(sT[kh][ks2 * 8 + 2 * j], sT[kh][ks2 * 8 + 2 * j + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const unsigned a = so + vbase + (8 * kh + 4 * ks2) * QUAD + db * 256;
                    const s16x4 v0 = lds_read_tr16(lds + a), v1 = lds_read_tr16(lds + a + 2 * QUAD);
                    const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    oT[db] = VL2_MFMA32(vf, pf, oT[db]);
                }
            }
    };


    // ---- main loop: tile t lives in stage t & 1.  Top of tile t: this wave's pieces of tile t have landed (its only
    //      outstanding VMEM), the barrier makes everyone's pieces visible AND proves every wave is done reading the other stage
    //      (tile t-1), which the DMA of tile t+1 may therefore overwrite while tile t is computed.  One barrier per tile.
    // a wave whose 32 query rows all lie past nq (the last query block of a 577-row ViT frame: rows 65..127 of it) still moves its
    // share of every K/V tile and meets every barrier, but skips the arithmetic: its issue slots go to the waves it shares a SIMD with
    const bool live = cls_w || q0 + wave * 32 < p.nq;
    const int nt_s = (ntiles - st + NS - 1) / NS;           // tiles of this stream; every wave meets the barriers of all `nsteps`
    const int nsteps = (ntiles + NS - 1) / NS;
    if (NS == 1 || nt_s > 0) dma_tile(0, 0);
    for (int t = 0; t < nsteps; t += 2) {
        VL2_WAIT_VMCNT(0);
        VL2_ATTN2_BARRIER();
        if (t + 1 < nt_s) dma_tile(t + 1, STAGE);
        if (live && (NS == 1 || t < nt_s)) compute_tile(t, 0);
        if (t + 1 >= nsteps) break;
        VL2_WAIT_VMCNT(0);
        VL2_ATTN2_BARRIER();
        if (t + 2 < nt_s) dma_tile(t + 2, 0);
        if (live && (NS == 1 || t + 1 < nt_s)) compute_tile(t + 1, STAGE);
    }

    if constexpr (NS == 2) {
        // merge of the two streams' online-softmax states: [4 row groups][NDB * 16 + 2 values][64 lanes] fp32 over the tile images
        constexpr int NR = NDB * 16 + 2;
        float* const xch = (float*)lds_all + (wave * NR) * 64 + lane;
        __syncthreads();                                     // every wave is done reading its last tile
        if (st == 1) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(db * 16 + r) * 64] = oT[db][r];
            xch[(NDB * 16) * 64] = m;
            xch[(NDB * 16 + 1) * 64] = l;
        }
        __syncthreads();
        if (st == 1) return;
        const float m1 = xch[(NDB * 16) * 64], l1 = xch[(NDB * 16 + 1) * 64];
        const float mn = fmaxf(m, m1);
        const float a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);     // an empty stream: m1 = -1e30, l1 = 0, O = 0
        l = __builtin_fmaf(l, a0, l1 * a1);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[db][r] = __builtin_fmaf(oT[db][r], a0, xch[(db * 16 + r) * 64] * a1);
    }

    if (qrow < p.nq) {
        const float inv = 1.0f / l;
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

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

template <int D, bool CAUSAL>
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
                    oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oT[db], 0, 0, 0);
                }
            }
    };

    // ---- main loop: tile t lives in stage t & 1.  Top of tile t: this wave's pieces of tile t have landed (its only
    //      outstanding VMEM), the barrier makes everyone's pieces visible AND proves every wave is done reading the other stage
    //      (tile t-1), which the DMA of tile t+1 may therefore overwrite while tile t is computed.  One barrier per tile.
    dma_tile(0, 0);
    for (int t = 0; t < ntiles; t += 2) {
        VL2_WAIT_VMCNT(0);
        VL2_ATTN2_BARRIER();
        if (t + 1 < ntiles) dma_tile(t + 1, STAGE);
        compute_tile(t, 0);
        if (t + 1 >= ntiles) break;
        VL2_WAIT_VMCNT(0);
        VL2_ATTN2_BARRIER();
        if (t + 2 < ntiles) dma_tile(t + 2, 0);
        compute_tile(t + 1, STAGE);
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

// ------------------------------------------------------------------------------------------------------------------
// attn_res64: the ViT shape (non-causal, D = 64, nk <= 640: CLIP 577, SigLIP-style grids up to 25 x 25) with the WHOLE K and V
// of a (frame, head) resident in LDS: 2 x 80 KiB = all 160 KiB of the CU.  One 8-wave workgroup per (frame, head[, q split]);
// every wave walks 32-row q blocks (block j of the workgroup -> wave j % 8) over the <= 10 resident KV tiles.
//   * K/V are read from HBM/L2 ONCE per (frame, head) (the tiled kernels re-stage them for each of the 5 q blocks) by LDS-DMA,
//     all tiles requested up front; a wave's FIRST block consumes them as they land (counted vmcnt + one barrier per tile),
//     its later blocks run with no barrier and no wait at all;
//   * the tile loop is software-pipelined inside the wave: QK^T of tile t+1 is issued before the softmax of tile t, so the
//     matrix pipe works through 8 MFMAs while the VALU does max / exp2 / sum / pack (two score sets live: 64 VGPRs, affordable
//     at D = 64), and the partner wave on the SIMD fills what is left;
//   * T = 16 frames x 16 heads = 256 workgroups = one per CU, perfectly balanced (qsplit spreads fewer (frame, head) pairs).
// Same LDS images and fragment maps as attn2_fwd_kernel<64> (tile t at byte t * 8192 of each image).
#define ATTN_RES_NT 10
__device__ __forceinline__ void attn_wait_vmcnt_rt(int n) {      // s_waitcnt takes an immediate: n = 2 * (tiles still in flight)
    switch (n) {
        case 0: VL2_WAIT_VMCNT(0); break;   case 2: VL2_WAIT_VMCNT(2); break;   case 4: VL2_WAIT_VMCNT(4); break;
        case 6: VL2_WAIT_VMCNT(6); break;   case 8: VL2_WAIT_VMCNT(8); break;   case 10: VL2_WAIT_VMCNT(10); break;
        case 12: VL2_WAIT_VMCNT(12); break; case 14: VL2_WAIT_VMCNT(14); break; case 16: VL2_WAIT_VMCNT(16); break;
        default: VL2_WAIT_VMCNT(18); break;
    }
}

__global__ __launch_bounds__(512, 2) void attn_res64_kernel(AttnArgs p) {
    constexpr int D = 64, NKS = 4, NDB = 2, TILE = 8192, IMG = ATTN_RES_NT * TILE, QUAD = 512;
    __shared__ __attribute__((aligned(16))) unsigned char lds_mem[2 * IMG];
    unsigned char* const lds = lds_mem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.x, b = blockIdx.y, qz = blockIdx.z, qsplit = gridDim.z;
    const int hk = h / p.group;
    const bf16_t* Q = p.q + b * p.q_bs + h * p.q_hs;
    const bf16_t* K = p.k + b * p.k_bs + hk * p.k_hs;
    const bf16_t* V = p.v + b * p.v_bs + hk * p.v_hs;
    const int ntiles = (p.nk + 63) >> 6;                       // <= ATTN_RES_NT (checked by the launcher)
    const int nqb = (p.nq + 31) >> 5;                          // 32-row q blocks of the sequence
    const int my_blocks = (nqb - qz + qsplit - 1) / qsplit;    // blocks qz, qz + qsplit, ... belong to this workgroup

    // ---- Q fragments of this wave's FIRST block, fetched and waited for BEFORE the LDS-DMA burst: the counted vmcnt of the
    //      first pass must see the tile requests only (and hipcc waits vmcnt(0) for an ordinary load issued beside LDS-DMA)
    bf16x8 qf[NKS];
    {
        const int qr = (qz + qsplit * wave) * 32 + l31;
        const int qr_c = qr < p.nq ? qr : p.nq - 1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(Q + (size_t)qr_c * p.q_rs + ks * 16 + hi * 8);
        VL2_WAIT_VMCNT(0);
    }
    // ---- request every tile now: per tile this wave issues K piece `wave` and V piece `wave` (1 KiB each)
    {
        const int slot = (wave << 6) + lane;
        const int R = slot >> 4, s = (slot & 15) ^ (R & 15);
        const unsigned koff = (unsigned)((2 * R + (s >> 3)) * p.k_rs + (s & 7) * 8) * 2u;
        const int blk = slot >> 3, w8 = slot & 7;
        const unsigned voff = (unsigned)(((blk >> 2) * 4 + (w8 >> 1)) * p.v_rs + (blk & 3) * 16 + (w8 & 1) * 8) * 2u;
        const int k_bytes = ((p.nk - 1) * p.k_rs + D) * 2, v_bytes = ((p.nk - 1) * p.v_rs + D) * 2;
        for (int t = 0; t < ntiles; ++t) {
            const int kskip = t * 64 * p.k_rs * 2, vskip = t * 64 * p.v_rs * 2;
            const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)K + kskip), 0, k_bytes > kskip ? k_bytes - kskip : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)V + vskip), 0, v_bytes > vskip ? v_bytes - vskip : 0, 0x00020000);
            lds_dma16(rsK, lds + t * TILE + (wave << 10), koff);
            lds_dma16(rsV, lds + IMG + t * TILE + (wave << 10), voff);
        }
    }
    unsigned kbase[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kbase[ks] = (unsigned)attn_k_off<D>(l31, ks * 2 + hi);
    const int g4 = lane >> 4, m16 = lane & 15;
    const unsigned vbase = (unsigned)(IMG + (g4 & 1) * 128 + (g4 >> 1) * QUAD + (m16 >> 2) * 32 + (m16 & 3) * 8);
    const float c = p.scale_log2e;

    for (int j = wave; j < my_blocks || j == wave; j += 8) {    // every wave runs the first (synchronising) pass once
        const bool have = j < my_blocks;
        const bool sync = j == wave;                            // first pass: tiles are still landing
        const int q0 = (qz + qsplit * j) * 32;
        const int qrow = q0 + l31;
        if (!sync) {                                            // later blocks: nothing else is in flight
            const int qrow_c = qrow < p.nq ? qrow : p.nq - 1;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(Q + (size_t)qrow_c * p.q_rs + ks * 16 + hi * 8);
        }
        f32x16 oT[NDB];
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
        float m = -1e30f, l = 0.f;

        auto qk = [&](int t) {                                  // S^T = K_t . Q^T  (8 MFMAs)
            if (sync) { attn_wait_vmcnt_rt(2 * (ntiles - 1 - t)); VL2_ATTN2_BARRIER(); }
            f32x16 s0, s1;
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 k0 = *(const bf16x8*)(lds + t * TILE + kbase[ks]);
                const bf16x8 k1 = *(const bf16x8*)(lds + t * TILE + 4096 + kbase[ks]);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], ks == 0 ? zero16 : s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], ks == 0 ? zero16 : s1, 0, 0, 0);
            }
            struct { f32x16 a, b; } r = {s0, s1};
            return r;
        };
        auto softmax_pv = [&](f32x16& s0, f32x16& s1, int t) {
            const int kv0 = t * 64;
            if (kv0 + 64 > p.nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s0[r] = key < p.nk ? s0[r] : -1e30f;
                    s1[r] = key + 32 < p.nk ? s1[r] : -1e30f;
                }
            }
            float mt = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, fmaxf(s0[r], s1[r]));
            mt = fmaxf(mt, __shfl_xor(mt, 32)) * c;
            constexpr float THR = 6.0f;
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
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c, -m));
                s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c, -m));
                rs += s0[r] + s1[r];
            }
            l += rs + __shfl_xor(rs, 32);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
                    const f32x16& sv = kh == 0 ? s0 : s1;
                    u32x4 pw;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) pw[jj] = pack2bf(sv[ks2 * 8 + 2 * jj], sv[ks2 * 8 + 2 * jj + 1]);
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const unsigned a = (unsigned)(t * TILE) + vbase + (8 * kh + 4 * ks2) * QUAD + db * 256;
                        const s16x4 v0 = lds_read_tr16(lds + a), v1 = lds_read_tr16(lds + a + 2 * QUAD);
                        const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oT[db], 0, 0, 0);
                    }
                }
        };
        // pipelined over the tiles, two named score sets (A: even tiles, B: odd tiles)
        auto sA = qk(0);
        for (int t = 0; t < ntiles; t += 2) {
            auto sB = sA;
            if (t + 1 < ntiles) sB = qk(t + 1);
            softmax_pv(sA.a, sA.b, t);
            if (t + 1 >= ntiles) break;
            if (t + 2 < ntiles) sA = qk(t + 2);
            softmax_pv(sB.a, sB.b, t + 1);
        }
        if (have && qrow < p.nq) {
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
}

// ------------------------------------------------------------------------------------------------------------------
// attn2p: the causal prefill shape (D = 128) is bound by its longest chain -- the last q block of S = 1621 walks 26 KV tiles, one
// dependent {QK^T -> softmax -> PV} sequence per tile on a wave that mostly has its SIMD to itself (416 workgroups on 256 CUs):
// ~3 k cycles per tile in attn2_fwd_kernel although the tile holds only 1 k cycles of MFMA work.  Here the sequence is
// software-pipelined inside the wave: QK^T of tile t+1 is issued BEFORE the softmax of tile t, so its 16 MFMAs run on the matrix
// pipe while the VALU does max / exp2 / sum / pack for tile t, then PV of tile t follows.  Cost: two live score sets
// (+32 VGPRs -> more than 256: one wave per SIMD, one workgroup per CU) and a THREE-stage K/V ring (tile t+2 is requested while
// tile t is still needed for its PV and tile t+1 for its QK^T): 96 KiB of LDS.
//   iteration t:  vmcnt(0) + barrier (tile t+1 landed for everyone; everyone is past PV(t-1), so stage (t+2)%3 is free)
//                 -> request tile t+2 -> QK^T(t+1) -> softmax(t) -> PV(t).
template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn2p_fwd_kernel(AttnArgs p) {
    constexpr int D = 128, NKS = 8, NDB = 4, K_BYTES = 64 * D * 2, STAGE = 2 * K_BYTES, PPW = 4, QUAD = (D / 16) * 128, KH_STEP = 32 * 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds_mem[3 * STAGE];
    unsigned char* const lds = lds_mem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
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
    const int qrow = q0 + wave * 32 + l31;
    const int qrow_c = qrow < p.nq ? qrow : p.nq - 1;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(Q + (size_t)qrow_c * p.q_rs + ks * 16 + hi * 8);
    VL2_WAIT_VMCNT(0);                          // the counted waits below must see LDS-DMA only
    int kmax = p.nk;
    if (CAUSAL) { const int lim = q0 + 128 + p.causal_off; kmax = lim < kmax ? lim : kmax; }
    const int ntiles = (kmax + 63) >> 6;

    unsigned koff[PPW], voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int slot = ((i * 4 + wave) << 6) + lane;
        const int row = slot >> 4, chunk = (slot & 15) ^ (row & 15);
        koff[i] = (unsigned)(row * p.k_rs + chunk * 8) * 2u;
        const int blk = slot >> 3, w8 = slot & 7;
        voff[i] = (unsigned)(((blk >> 3) * 4 + (w8 >> 1)) * p.v_rs + (blk & 7) * 16 + (w8 & 1) * 8) * 2u;
    }
    const int k_bytes = ((p.nk - 1) * p.k_rs + D) * 2, v_bytes = ((p.nk - 1) * p.v_rs + D) * 2;
    auto dma_tile = [&](int t, unsigned so) {
        const int kskip = t * 64 * p.k_rs * 2, vskip = t * 64 * p.v_rs * 2;
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)K + kskip), 0, k_bytes > kskip ? k_bytes - kskip : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)V + vskip), 0, v_bytes > vskip ? v_bytes - vskip : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < PPW; ++i) lds_dma16(rsK, lds + so + ((i * 4 + wave) << 10), koff[i]);
#pragma unroll
        for (int i = 0; i < PPW; ++i) lds_dma16(rsV, lds + so + K_BYTES + ((i * 4 + wave) << 10), voff[i]);
    };
    unsigned kbase[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kbase[ks] = (unsigned)attn_k_off<D>(l31, ks * 2 + hi);
    const int g4 = lane >> 4, m16 = lane & 15;
    const unsigned vbase = (unsigned)(K_BYTES + (g4 & 1) * 128 + (g4 >> 1) * QUAD + (m16 >> 2) * 32 + (m16 & 3) * 8);

    f32x16 oT[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const float c = p.scale_log2e;

    auto qk = [&](f32x16& s0, f32x16& s1, unsigned so) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 k0 = *(const bf16x8*)(lds + so + kbase[ks]);
            const bf16x8 k1 = *(const bf16x8*)(lds + so + KH_STEP + kbase[ks]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], ks == 0 ? zero16 : s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], ks == 0 ? zero16 : s1, 0, 0, 0);
        }
    };
    auto softmax_pv = [&](f32x16& s0, f32x16& s1, int t, unsigned so) {
        const int kv0 = t * 64;
        const int wq0 = q0 + wave * 32;
        if ((kv0 + 64 > p.nk) || (CAUSAL && (kv0 + 63 > wq0 + p.causal_off))) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                s0[r] = (key < p.nk && (!CAUSAL || key <= qrow + p.causal_off)) ? s0[r] : -1e30f;
                s1[r] = (key + 32 < p.nk && (!CAUSAL || key + 32 <= qrow + p.causal_off)) ? s1[r] : -1e30f;
            }
        }
        float mt = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, fmaxf(s0[r], s1[r]));
        mt = fmaxf(mt, __shfl_xor(mt, 32)) * c;
        constexpr float THR = 6.0f;
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
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c, -m));
            s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c, -m));
            rs += s0[r] + s1[r];
        }
        l += rs + __shfl_xor(rs, 32);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const f32x16& sv = kh == 0 ? s0 : s1;
                u32x4 pw;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) pw[jj] = pack2bf(sv[ks2 * 8 + 2 * jj], sv[ks2 * 8 + 2 * jj + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const unsigned a = so + vbase + (8 * kh + 4 * ks2) * QUAD + db * 256;
                    const s16x4 v0 = lds_read_tr16(lds + a), v1 = lds_read_tr16(lds + a + 2 * QUAD);
                    const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oT[db], 0, 0, 0);
                }
            }
    };
    // one pipelined step: scores of tile t are in (c0, c1); produces the scores of tile t+1 in (n0, n1).  S0/S1/S2 = byte offsets
    // of the stages holding tiles t, t+1, t+2
    auto step = [&](f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1, int t, unsigned S0, unsigned S1, unsigned S2) {
        VL2_WAIT_VMCNT(0);
        VL2_ATTN2_BARRIER();
        if (t + 2 < ntiles) dma_tile(t + 2, S2);
        if (t + 1 < ntiles) qk(n0, n1, S1);
        softmax_pv(c0, c1, t, S0);
    };
    f32x16 a0, a1, b0, b1;
    dma_tile(0, 0);
    if (ntiles > 1) dma_tile(1, STAGE);
    VL2_WAIT_VMCNT(0);
    VL2_ATTN2_BARRIER();
    qk(a0, a1, 0);
    // Stage offsets rotate at run time (three adds per tile) and the new scores are copied into the current set (32 moves per
    // tile): the fully static form (stages x score sets = a 6-step unrolled body) made hipcc spill 836 bytes per lane.
    unsigned S0 = 0, S1 = STAGE, S2 = 2 * STAGE;
    for (int t = 0; t < ntiles; ++t) {
        step(a0, a1, b0, b1, t, S0, S1, S2);
        a0 = b0;
        a1 = b1;
        const unsigned r = S0;
        S0 = S1;
        S1 = S2;
        S2 = r;
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

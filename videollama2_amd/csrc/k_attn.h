// Fused (flash-style) attention forward for gfx950, used for
//   * CLIP-ViT MHSA, non-causal, B = frames, H = 16, N = 577, D = 64  (HF:models/clip/modeling_clip.py
//     eager_attention_forward: softmax(QK^T * d^-0.5) V, softmax in fp32; the reference forces flash-attn,
//     videollama2/model/encoder.py:24)
//   * Mistral causal GQA prefill, H = 32, KV = 8, D = 128               (HF:models/mistral/modeling_mistral.py
//     eager_attention_forward + repeat_kv; causal mask from the cache length)
//
// Transposed formulation so every softmax statistic is lane-local (guide App. B "swapped QK^T"):
//   S^T[key][q] = K . Q^T   (A = K rows from LDS, B = Q^T from registers)  -> lane (q = lane&31) owns 32 of the
//   64 scores of its q row per KV tile, its partner lane^32 the other 32;
//   O^T[d][q]  += V^T . P^T (A = V^T from a transposed LDS image, B = P^T straight from the score registers: the
//   MFMA contraction order over keys is a free permutation, so the B fragment is exactly the registers a lane
//   already owns and the V^T fragment is read with the same permutation).
// 4 waves x 32 q rows per workgroup, KV tile = 64 keys, K/V staged global -> registers -> LDS (next tile's loads
// are issued before the current tile's MFMAs), online softmax in the exp2 domain.
// NG = 2 (causal prefill of ONE sequence: S*H/32 = 1664 waves for 1024 SIMDs, and the longest q block is a chain of 26
// dependent tiles that mostly has its SIMD to itself): a second group of 4 waves takes every other KV tile of the same
// 128 q rows with its own K/V staging buffers and its own running (m, l, O); the two partial results are merged through
// LDS at the end (flash-decoding inside the workgroup).  Two independent chains per SIMD, half the chain length.
#pragma once
#include "dev_common.h"

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    long q_bs, q_hs; int q_rs;      // element strides: batch, head, row
    long k_bs, k_hs; int k_rs;
    long v_bs, v_hs; int v_rs;
    long o_bs, o_hs; int o_rs;
    int nq, nk, group;              // group = q heads per kv head
    int heads, batch;
    float scale_log2e;              // softmax scale * log2(e)
    int causal_off;                 // key j visible to q row i iff j <= i + causal_off
};

template <int D> __device__ __forceinline__ int attn_k_off(int row, int chunk);
template <> __device__ __forceinline__ int attn_k_off<128>(int row, int chunk) { return ((row << 4) + (chunk ^ (row & 15))) << 4; }
// D = 96 (SigLIP-so400m's head_dim 72 padded to the next multiple of 32): the K image keeps the 256-B row pitch of D = 128,
// only 12 of the 16 chunks of a row are filled and read
template <> __device__ __forceinline__ int attn_k_off<96>(int row, int chunk) { return attn_k_off<128>(row, chunk); }
template <> __device__ __forceinline__ int attn_k_off<64>(int row, int chunk) {
    const int R = row >> 1, s = ((row & 1) << 3) | chunk;
    return ((R << 4) + (s ^ (R & 15))) << 4;
}
// V^T image: row d = 64 keys * 2 B = 128 B = 8 chunks of 16 B.  Chunk (kb, hi) (kb = 16-key block 0..3) holds, in
// order, keys 16kb + 4hi + {0,1,2,3, 8,9,10,11}: exactly the 8 keys lane-half `hi` contracts in one PV MFMA, so the
// A fragment is ONE ds_read_b128.  Rows are swizzled like the K tile of the 64-wide case (two rows per bank row).
template <int D>
__device__ __forceinline__ int attn_vt_off(int d, int c16) {
    // slot key: R & 15 alone keeps the fragment READ conflict-free (16 consecutive R per instruction), but the transposing 4-byte
    // WRITES touch rows d, d+8, ... = R, R+4, ..., R+16, ... in one instruction and R / R+16 (+32, +48) then share banks (PMC: 32 %
    // / 53 % of LDS-active cycles were conflicts at D = 64 / 128).  The high bits of R are folded into key bits the write
    // pattern leaves free: bit 1 at D = 64 (a wave writes 8 key pairs: chunk bit 0 varies), bits 0-1 at D >= 96.
    const int R = d >> 1, s = ((d & 1) << 3) | c16;
    const int key = D == 64 ? (R & 15) ^ ((R >> 4) << 1) : (R ^ (R >> 4)) & 15;
    return ((R << 4) + (s ^ key)) << 4;
}

template <int D, bool CAUSAL, int NG = 1>
__global__ __launch_bounds__(256 * NG, 2) void attn_fwd_kernel(AttnArgs p) {
    constexpr int KCH = D / 8;                 // 16-B chunks per K row
    constexpr int KPT = 64 * KCH / 256;        // K chunks per thread per tile
    constexpr int VITEMS = 32 * KCH;           // V (key-pair, chunk) items per tile
    constexpr int VPT = (VITEMS + 255) / 256;  // ... per thread (D = 96: 1.5 -> 2 passes, the second half-populated)
    constexpr int NKS = D / 16;                // k-steps of the QK^T MFMA chain
    constexpr int NDB = D / 32;                // 32-row d blocks of O^T
    constexpr int KS_BYTES = 64 * (D == 96 ? 128 : D) * 2, VT_BYTES = D * 128;
    static_assert(NG == 1 || (NG == 2 && NDB % 2 == 0), "KV groups: 1, or 2 with an even number of O^T blocks");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NG * (KS_BYTES + VT_BYTES)];

    // tid / wave: within the KV group (all staging and fragment indexing is per group)
    const int tid = threadIdx.x & 255, grp = NG > 1 ? (int)threadIdx.x >> 8 : 0, lane = tid & 63, wave = tid >> 6;
    unsigned char* const Ks = lds + grp * (KS_BYTES + VT_BYTES);
    unsigned char* const Vt = Ks + KS_BYTES;
    const int hi = lane >> 5, l31 = lane & 31;
    // Non-causal: grid = (q blocks, heads, batch).  Causal: 1-D grid over (q block, head*batch) with the heads fastest, and
    // q block b costs b+1 KV tiles: the first ceil(256 / (heads*batch)) q-block ranks -- one workgroup per CU -- are the
    // LONGEST blocks in descending order, the rest follow in ASCENDING order, so that the second workgroup a CU receives is
    // short where the first one is long (the longest block then has the CU almost to itself instead of sharing it with an
    // arbitrary partner: per-CU totals 13-14 tiles instead of up to 26 at S = 1621).  Pure scheduling hint, any
    // placement is correct.
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

    // K/V tile loads as raw buffer loads: per-thread byte offsets inside a tile are loop-invariant VGPRs, the tile's position
    // goes into the (scalar) buffer base, and rows past nk fall outside NUM_RECORDS and read as 0 (their scores are masked,
    // their P is 0) -- no per-tile address arithmetic (it was 8 64-bit multiply-adds + clamps per tile and thread).
    u32x4 kreg[KPT], vreg[VPT][2];
    int koff[KPT], voff[VPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int c = tid + 256 * i, row = c / KCH, ch = c % KCH;
        koff[i] = (row * p.k_rs + ch * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = tid + 256 * i, kp = c / KCH, ch = c % KCH;
        voff[i] = (2 * kp * p.v_rs + ch * 8) * 2;
    }
    const int k_bytes = ((p.nk - 1) * p.k_rs + D) * 2, v_bytes = ((p.nk - 1) * p.v_rs + D) * 2;   // valid bytes behind K / V
    auto load_tile = [&](int kv0) {
        const int kskip = kv0 * p.k_rs * 2, vskip = kv0 * p.v_rs * 2;
        const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)K + kskip), 0, k_bytes > kskip ? k_bytes - kskip : 0, 0x00020000);
        const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)V + vskip), 0, v_bytes > vskip ? v_bytes - vskip : 0, 0x00020000);
        const int vrow = p.v_rs * 2;
        const auto rsV1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)V + vskip + vrow), 0,
                                                            v_bytes > vskip + vrow ? v_bytes - vskip - vrow : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < KPT; ++i) kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, koff[i], 0, 0));
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            if (VITEMS % 256 != 0 && tid + 256 * i >= VITEMS) continue;
            vreg[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, voff[i], 0, 0));
            vreg[i][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV1, voff[i], 0, 0));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int c = tid + 256 * i, row = c / KCH, ch = c % KCH;
            *(u32x4*)(Ks + attn_k_off<D>(row, ch)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = tid + 256 * i, kp = c / KCH, ch = c % KCH;
            if (VITEMS % 256 != 0 && c >= VITEMS) continue;
            // keys 2kp, 2kp+1 -> chunk (kb = key>>4, hi = bit 2 of key), element (bit 3 of key)*4 + (key & 3)
            const int k16 = (2 * kp) & 15;
            const int c16 = ((2 * kp) >> 4) * 2 + ((k16 >> 2) & 1);
            const int eoff = ((k16 >> 3) * 4 + (k16 & 3)) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned a = (vreg[i][0][j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                const unsigned bb = (vreg[i][1][j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                *(unsigned*)(Vt + attn_vt_off<D>(ch * 8 + j, c16) + eoff) = a | (bb << 16);
            }
        }
    };

    f32x16 oT[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
    float m = -1e30f, l = 0.f;      // m: running max in the exp2 domain (first tile always rescales: mt - m is huge)

    load_tile(grp * 64);                      // rows are clamped: harmless for a group without a tile
    for (int t0 = 0; t0 < ntiles; t0 += NG) {
        const int t = t0 + grp, kv0 = t * 64;
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t + NG < ntiles) load_tile(kv0 + 64 * NG);
        if (NG > 1 && t >= ntiles) continue;   // odd tile count: the last round has work for group 0 only (barriers are above)

        // S^T = K . Q^T
        f32x16 sT[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(Ks + attn_k_off<D>(kh * 32 + l31, ks * 2 + hi));
                sT[kh] = VL2_MFMA32(kf, qf[ks], ks == 0 ? zero16 : sT[kh]);
            }
        }
        // online softmax (exp2 domain: p = 2^(s*c - m), c = scale*log2 e folded into one FMA per score).
        // lane owns keys kv0 + 32kh + (r&3) + 8(r>>2) + 4hi of row qrow.
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
        // deferred rescale (guide T13): keep the old running max while the tile max exceeds it by < 2^THR; P is then
        // bounded by 2^THR (fp32 accumulators, bf16 P: fine).  THR = 0 would rescale every tile.
        constexpr float THR = 6.0f;
        if (!__all(mt - m <= THR)) {
            // floor at -1e28: a row that has met only masked keys so far (possible for the second KV group, whose first tile can
            // lie entirely above the row's diagonal) keeps p = 2^(-1e30 c + 1e28) = 0 instead of 2^(rounding residue of -1e30 c)
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
                const float pv = __builtin_amdgcn_exp2f(fmaf(sT[kh][r], c, -m));   // bare v_exp_f32: inputs <= THR, tiny results flush to 0
                sT[kh][r] = pv;
                rs += pv;
            }
        l += rs + __shfl_xor(rs, 32);

        // O^T += V^T . P^T
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                u32x4 pw;
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[j] = pack2bf(sT[kh][ks2 * 8 + 2 * j], sT[kh][ks2 * 8 + 2 * j + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
                const int c16 = (kh * 2 + ks2) * 2 + hi;
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const bf16x8 vf = *(const bf16x8*)(Vt + attn_vt_off<D>(db * 32 + l31, c16));
                    oT[db] = VL2_MFMA32(vf, pf, oT[db]);
                }
            }
    }

    if constexpr (NG > 1) {
        // group 1's (m, l, O^T) -> LDS -> group 0, two O^T blocks per round (32 KiB) + 2 KiB of statistics.  Lane-major
        // 16-B slots: conflict-free both ways.  A group that saw no tile carries m = -1e30, l = 0, O = 0 -> weight 0.
        float* const xo = (float*)lds;
        float* const xs = (float*)(lds + 32768);
        float a0 = 1.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < NDB / 2; ++r) {
            __syncthreads();                                   // tile reads (r = 0) / the previous round's reads are done
            if (grp == 1) {
                if (r == 0) { xs[(wave * 64 + lane) * 2] = m; xs[(wave * 64 + lane) * 2 + 1] = l; }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 w = {oT[2 * r + i][4 * q4], oT[2 * r + i][4 * q4 + 1], oT[2 * r + i][4 * q4 + 2], oT[2 * r + i][4 * q4 + 3]};
                        *(f32x4*)(xo + (((wave * 2 + i) * 4 + q4) * 64 + lane) * 4) = w;
                    }
            }
            __syncthreads();
            if (grp == 0) {
                if (r == 0) {
                    const float m1 = xs[(wave * 64 + lane) * 2], l1 = xs[(wave * 64 + lane) * 2 + 1];
                    const float ms = fmaxf(m, m1);
                    a0 = __builtin_amdgcn_exp2f(m - ms);
                    a1 = __builtin_amdgcn_exp2f(m1 - ms);
                    l = a0 * l + a1 * l1;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 w = *(const f32x4*)(xo + (((wave * 2 + i) * 4 + q4) * 64 + lane) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) oT[2 * r + i][4 * q4 + j] = a0 * oT[2 * r + i][4 * q4 + j] + a1 * w[j];
                    }
            }
        }
        if (grp != 0) return;
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

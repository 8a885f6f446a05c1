// gemm6: the PERSISTENT form of the 256-column ping-pong GEMM (gemm4, k_gemm.h) for bf16 outputs without a residual
// (ViT q/k/v and fc1 + QuickGELU, the STC 1x1 convolutions, the decoder's q/k/v and gate/up + SwiGLU).
//
// Why (profiles/r03_experiments.md section 2): one 128 KiB-LDS workgroup owns a CU, so between two tiles of a
// CU nothing overlaps -- the C^T epilogue (7-12 us per 256 x 256 tile = 22-36 % of a K = 1024 GEMM), the drain of its stores, the
// launch of the next workgroup and its ring fill are all exposed.  Here ONE workgroup per CU walks its tiles (b, b + G, b + 2G, ...):
//   * the LDS ring never drains: the LDS-DMA issue pointer runs three slabs ahead of the MFMA pointer ACROSS tile boundaries;
//   * the per-tile epilogue vectors (bias, LayerNorm column sums of W', the rows' (mean, rstd)) arrive by LDS-DMA with the
//     tile's first slab (one extra 1 KiB piece on waves 0-3), so the epilogue issues no global load -- stores and LDS-DMA share
//     vmcnt on gfx9, loads retire in order, and a counted wait only has to count LOADS issued later (pending stores can only
//     make it stricter, never wrong);
//   * DB = true (192-row tiles: 96 accumulator registers per wave): TWO accumulator sets; tile i's epilogue is drained one
//     32 x 32 block per MFMA phase under the first phases of tile i + 1 -- VALU work in the shadow of the wave's own MFMAs,
//     stores never waited for;
//   * DB = false (256-row tiles: 128 accumulators, no room for a second set): the epilogue stays between the tiles, both wave
//     groups in the same barrier interval, but launch, ring fill and store drain are gone.
//   * tiles are handed out DYNAMICALLY when the call carries a counter block (GemmArgs.tile_ctr, DB = false): a workgroup's first tile
//     is static (tile b), every further one is G + atomicAdd(counter, 1).  A static b, b + G, b + 2G ... walk makes the kernel as slow as
//     its slowest CU -- measured: on one box of the pool the static form lost 1 ms per ViT pass in the pipeline (8.55 vs 7.58 ms) while
//     winning 10 % per kernel on three other boxes (XCDs clock independently; one slow or late workgroup holds two tiles back).  The
//     atomic for the NEXT tile is issued by one lane in the tile's first phase, parked in LDS one phase later (the counted LDS-DMA waits
//     of the ring cover it: loads retire in order) and read by every wave three phases before the tile ends; the last workgroup to
//     finish re-arms the block for the next launch.
// Same slabs, same k order, same epilogue arithmetic as gemm4 + gemm_store_tr -> the same bits per element.
#pragma once
#include "k_gemm.h"

#define GEMM6_AUX_OFF (4 * GEMM4_STAGE)                       // behind the 4-stage ring: [tile parity][bias | colsum | (mean, rstd)]
#define GEMM6_AUX_BYTES 4096                                  // 1 KiB bias (256 f32) + 1 KiB colsum + 2 KiB row table (256 x 8 B)
#define GEMM6_SLOT_OFF (GEMM6_AUX_OFF + 2 * GEMM6_AUX_BYTES)   // one word: the next tile of this workgroup (dynamic form)
#define GEMM6_LDS_BYTES (GEMM6_SLOT_OFF + 16)

// compile-time loop / constant helpers (no <utility>: the CPU test build compiles the kernel headers without the HIP headers)
template <int V> struct vl2_ic { static constexpr int value = V; };
template <int I, int N, typename F>
__device__ __forceinline__ void vl2_static_for(F&& f) {
    if constexpr (I < N) {
        f(vl2_ic<I>{});
        vl2_static_for<I + 1, N>(f);
    }
}

// lane id recomputed where it is needed (2 VALU) instead of kept in a register across the kernel; the opaque copy keeps the optimiser
// from hoisting what is derived from it (with two accumulator sets every hoisted loop-invariant is a spill)
#ifndef VL2_LANE_ID_FRESH
#define VL2_LANE_ID_FRESH(ln) do { ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(ln)); } while (0)
#endif

// dst = atomicAdd(ptr, 1) at agent scope (`global_atomic_add ... sc0`: what hipcc emits for __hip_atomic_fetch_add(..., AGENT)) WITHOUT
// a wait: the caller retires it with the counted vmcnt waits of its LDS-DMA ring (see gemm6_body).  The CPU test build defines its own.
#ifndef VL2_ATOMIC_INC_ASYNC
#define VL2_ATOMIC_INC_ASYNC(dst, ptr) do { const unsigned one_ = 1u; asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(dst) : "v"(ptr), "v"(one_) : "memory"); } while (0)
#endif

// one LDS word at byte offset `off` of the dynamic LDS array, as ds_write_b32 / ds_read_b32 (a volatile generic pointer would become a
// FLAT access followed by s_waitcnt vmcnt(0), which drains the LDS-DMA ring)
#ifndef VL2_LDS_I32
#define VL2_LDS_I32(off) (*(__attribute__((address_space(3))) int*)(vl2_smem + (off)))
#endif

// s_waitcnt vmcnt(n) for a wave-uniform runtime n in [0, 10]
__device__ __forceinline__ void gemm6_wait_vm(int n) {
    switch (n) {
        case 0: VL2_WAIT_VMCNT(0); break;
        case 1: VL2_WAIT_VMCNT(1); break;
        case 2: VL2_WAIT_VMCNT(2); break;
        case 3: VL2_WAIT_VMCNT(3); break;
        case 4: VL2_WAIT_VMCNT(4); break;
        case 5: VL2_WAIT_VMCNT(5); break;
        case 6: VL2_WAIT_VMCNT(6); break;
        case 7: VL2_WAIT_VMCNT(7); break;
        case 8: VL2_WAIT_VMCNT(8); break;
        case 9: VL2_WAIT_VMCNT(9); break;
        default: VL2_WAIT_VMCNT(10); break;
    }
}

// four v_permlane32_swap-able registers: (pk[0], pk[2]), (pk[1], pk[3]) as (vdst, src), see VL2_PERMLANE32_SWAP_8; s_nop 1 in front =
// the wait states a VALU write needs before the swap reads it, s_nop 1 behind = the same for the swap's results (the hazard
// recogniser does not see inside the asm)
#ifndef VL2_PERMLANE32_SWAP_4
#define VL2_PERMLANE32_SWAP_4(pk)                                                                                              \
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1"                         \
                 : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]))
#endif

// One epilogue CHUNK = 16 output columns x 32 rows of a wave's C^T accumulators: the 8 registers [8 gh, 8 gh + 8) of one 32 x 32
// block (SwiGLU: of the gate block a0 and the up block a1 -> silu(gate) * up; otherwise of block a0).  Arithmetic = gemm_store_tr's,
// element for element ((acc [norm]) + bias -> activation -> bf16); the per-column / per-row vectors come from the tile's LDS aux
// block instead of global memory; ONE 16-B store per lane.
//   m: global row of this lane (m0 + wave row + 32 mi + (lane & 31)); rit: its offset inside the tile
//   ct: first (un-halved) tile column of the block (SwiGLU: of the gate block); n0: first GEMM column of the tile
template <int ACT, bool SWIGLU, int GH>
__device__ __forceinline__ void gemm6_epi_chunk(const GemmArgs& p, const f32x16& a0, const f32x16& a1, int m, int rit,
                                                int ct, int n0, int hi, const unsigned char* aux) {
#pragma clang fp reassociate(off)
    const bool has_bias = !SWIGLU && p.bias != nullptr;
    const bool norm_rms = p.norm == 1;
    const bool norm_ln = !SWIGLU && p.norm == 2;
    const bool live = m < p.M;
    float mu = 0.f, rs = 1.f;
    if (p.norm) {
        const f32x2 st = *(const f32x2*)(aux + 2048 + rit * 8);
        mu = st[0];
        rs = st[1];
    }
    const int cb = ct + 4 * hi + 16 * GH;                                    // this lane's first tile column of register group g = 2 GH
    float x[8];
    if (SWIGLU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float g = a0[8 * GH + r], u = a1[8 * GH + r];
            if (norm_rms) { g *= rs; u *= rs; }
            x[r] = silu_f(g) * u;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = a0[8 * GH + r];
        if (norm_ln) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4 cs = *(const f32x4*)(aux + 1024 + (cb + 8 * g) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[4 * g + c] = __builtin_fmaf(-mu, cs[c], x[4 * g + c]) * rs;
            }
        } else if (norm_rms) {
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] *= rs;
        }
        if (has_bias) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4 bv = *(const f32x4*)(aux + (cb + 8 * g) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[4 * g + c] += bv[c];
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (ACT == ACT_QGELU) x[r] = quick_gelu_f(x[r]);
            if (ACT == ACT_GELU) x[r] = gelu_erf_f(x[r]);
            if (ACT == ACT_SILU) x[r] = silu_f(x[r]);
            if (ACT == ACT_GELU_TANH) x[r] = gelu_tanh_f(x[r]);
        }
    }
    uint32_t pk[4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        pk[2 * g] = pack2bf(x[4 * g], x[4 * g + 1]);
        pk[2 * g + 1] = pack2bf(x[4 * g + 2], x[4 * g + 3]);
    }
    VL2_PERMLANE32_SWAP_4(pk);
    bf16_t* crow = (bf16_t*)p.C + (size_t)m * p.ldc;
    const int oc = (SWIGLU ? ((n0 + ct) >> 1) : n0 + ct) + 16 * GH + 8 * hi;  // first of this lane's 8 output columns
    const u32x4 packed = {pk[0], pk[1], pk[2], pk[3]};
    if (live) *(u32x4*)(crow + oc) = packed;
}

// BM = 256 (DB must be false) or 192 (DB either way).  Grid: G <= tiles_m * tiles_n workgroups of 512 threads, one per CU.
// Requires: M >= BM (DB), bf16 output, no residual, no stats_out, norm via p.row_norm (or none), K % 32 == 0, K >= 32 * (NCH + 4)
// (DB: the chunks ride on the first NCH phases of the next tile), N % 256 == 0.
//
// Per tile: [phase 0, peeled] [DB: phases 1 .. NCH - 1, peeled, one epilogue chunk of the previous tile each] [hot loop up to phase
// nt - 4: slab t + 3 of the SAME tile is issued -- the loop body is gemm4's, no tile logic] [three peeled tail phases: slabs 0, 1, 2 of
// the NEXT tile are issued (its offsets and its aux block are set up in the first of them), or nothing behind the last tile].
// Counted waits: slab t + 1 must have landed, the two newer slabs stay in flight = 2 P loads of this wave (P = 4, or 3 for the waves
// with one A piece).  The aux piece (waves 0-3, with a tile's slab 0) is not counted: the wait is then one load stricter, never wrong.
template <int ACT, bool SWIGLU, int BM, bool DB>
__device__ __forceinline__ void gemm6_body(const GemmArgs& p) {
    static_assert(BM == 256 || BM == 192, "gemm6: 256- or 192-row tiles");
    static_assert(!(DB && BM == 256), "two accumulator sets only fit the 192-row geometry");
    constexpr int GR = BM / 2;
    constexpr int MI = BM == 256 ? 2 : 3, NJ = BM == 256 ? 4 : 2;
    constexpr int APG = GR / 16;
    constexpr int NSET = DB ? 2 : 1;
    constexpr int NCH = 2 * (SWIGLU ? MI * NJ / 2 : MI * NJ);               // epilogue chunks (16 output columns x 32 rows) per wave
    constexpr int NPEEL = DB ? NCH : 2;                                     // peeled head phases of a tile
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wrow = BM == 256 ? grp * 128 + (w4 >> 1) * 64 : grp * 96;
    const int wcol = BM == 256 ? (w4 & 1) * 128 : w4 * 64;

    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nt = p.K / GEMM4_BK;
    unsigned* const ctr = DB ? nullptr : p.tile_ctr;                        // [0] tiles handed out beyond the first G, [1] workgroups finished
    const bool dyn = ctr != nullptr;

    // logical tile t0 -> (m0, n0): gemm4's raster (4 tile-rows deep)
    auto tile_origin = [&](int t0, int& m0, int& n0) {
        const int grp_sz = 4 * p.tiles_n;
        const int first_m = (t0 / grp_sz) * 4;
        const int gm = (p.tiles_m - first_m) < 4 ? (p.tiles_m - first_m) : 4;
        m0 = (first_m + (t0 % grp_sz) % gm) * BM;
        if constexpr (DB) m0 = m0 + BM > p.M ? p.M - BM : m0;                // DB: the last row tile is shifted back to END at M (see below)
        n0 = ((t0 % grp_sz) / gm) * GEMM4_BN;
    };
    // static walk: tile `ord` of this workgroup = round `ord` of G tiles, XCD-aware order inside the round; >= ntiles: none
    auto static_t0 = [&](int ord) {
        const int base = ord * G;
        if (base + b >= ntiles) return ntiles;
        const int cnt = (ntiles - base) < G ? (ntiles - base) : G;
        return base + xcd_remap(b, cnt);
    };

    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    // aux pieces (waves 0-3, one each, with a tile's first slab): bias | colsum | row table rows [0,128) | rows [128,256).  A vector the
    // call does not carry is fetched from W instead (never read back); rows past M read as zeros through the record count.
    const bool has_aux = wave < 4;
    const void* aux_ptr = wave == 0 ? (const void*)(p.bias ? p.bias : (const float*)p.W)
                        : wave == 1 ? (const void*)(p.w_colsum ? p.w_colsum : (const float*)p.W)
                                    : (const void*)(p.row_norm ? p.row_norm : (const float*)p.W);
    const int aux_rec = wave < 2 ? p.N * 4 : (p.row_norm ? p.M * 8 : 0);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)aux_ptr, 0, aux_rec, 0x00020000);

    const bool two_a = BM == 256 || w4 < 2;
    // W: ONE per-lane byte offset; a wave's second piece is 64 rows further (the same swizzled chunk) = a uniform delta in the scalar offset.
    // A, DB = false: two per-lane offsets, rows past M clamp to M - 1 as in gemm4 (their outputs are never stored).
    // A, DB = true: no register for a second offset, so the delta form -- which rules out the per-lane clamp (and the hardware's range
    // check does not cover the scalar offset): the last row tile is shifted back to rows [M - BM, M) instead; every row a tile touches
    // exists, the rows it shares with its neighbour are computed -- and stored, with the same bits -- twice.  Requires M >= BM.
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const unsigned a_step = 128u * (unsigned)p.lda, w_step = 128u * (unsigned)p.ldw;   // bytes per 64 rows
    unsigned a_vo, a_vo1 = 0, w_vo;
    // offsets of tile (m0, n0) for the issue side
    auto set_offsets = [&](int m0, int n0) {
        int ln;                                                             // recomputed per tile, not kept in registers
        VL2_LANE_ID_FRESH(ln);
        const int aslot = ((grp * APG + w4) << 6) + ln, wslot = grp * 512 + (w4 << 6) + ln;
        const int Ra = aslot >> 4, spa = aslot & 15, Rw = wslot >> 4, spw = wslot & 15;
        const int rowa = 4 * Ra + (spa >> 2), chka = (spa & 3) ^ (Ra & 3);
        const int roww = 4 * Rw + (spw >> 2), chkw = (spw & 3) ^ (Rw & 3);
        if constexpr (DB) {
            a_vo = ((unsigned)(m0 + rowa) * (unsigned)p.lda + chka * 8) * 2;
        } else {
            int am = m0 + rowa, am1 = m0 + rowa + 64;
            am = am < p.M ? am : p.M - 1;
            am1 = am1 < p.M ? am1 : p.M - 1;
            a_vo = ((unsigned)am * (unsigned)p.lda + chka * 8) * 2;
            a_vo1 = ((unsigned)am1 * (unsigned)p.lda + chka * 8) * 2;
        }
        w_vo = ((unsigned)(n0 + roww) * (unsigned)p.ldw + chkw * 8) * 2;
    };
    // this wave's pieces of slab s (of the tile the offsets point at) into ring stage (g & 3)
    auto issue_slab = [&](unsigned g, int s) {
        const unsigned st = (g & 3u) * GEMM4_STAGE, kb = (unsigned)s * (GEMM4_BK * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * APG + w4) << 10)),
                                                 16, a_vo, kb, 0, 0);
        if (two_a)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * APG + 4 + w4) << 10)),
                                                     16, DB ? a_vo : a_vo1, DB ? kb + a_step : kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((grp * 8 + i * 4 + w4) << 10)),
                                                     16, w_vo, kb + i * w_step, 0, 0);
    };
    auto issue_aux = [&](int par, int m0, int n0) {
        if (has_aux) {
            const unsigned so = wave < 2 ? (unsigned)n0 * 4u : (unsigned)m0 * 8u + (wave == 3 ? 1024u : 0u);
            int ln;
            VL2_LANE_ID_FRESH(ln);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(vl2_smem + GEMM6_AUX_OFF + par * GEMM6_AUX_BYTES + (wave << 10)),
                                                     16, (unsigned)ln * 16u, so, 0, 0);
        }
    };
    // `slabs` newer slabs of this wave's LDS-DMA may stay in flight
    auto wait_dma = [&](int slabs) {
        if (slabs >= 2) { if (two_a) VL2_WAIT_VMCNT(8); else VL2_WAIT_VMCNT(6); }
        else if (slabs == 1) { if (two_a) VL2_WAIT_VMCNT(4); else VL2_WAIT_VMCNT(3); }
        else VL2_WAIT_VMCNT(0);
    };

    f32x16 acc[NSET][MI][NJ];
    bf16x8 fa[2][MI], fb[2][NJ];
    const int frow = lane & 31, fchk = lane >> 5;
    unsigned a_rd[2], b_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_rd[ks] = gemm4_lds_off(wrow + frow, ks * 2 + fchk);
        b_rd[ks] = 16384 + gemm4_lds_off(wcol + frow, ks * 2 + fchk);
    }

    // epilogue chunk c (16 output columns x 32 rows) of accumulator set S for the tile at (m0, n0) whose aux block has parity `par`
    auto epi_chunk = [&](auto S_, auto c_, int m0, int n0, int par) {
        constexpr int S = decltype(S_)::value, c = decltype(c_)::value, u = c >> 1, gh = c & 1;
        const unsigned char* aux = vl2_smem + GEMM6_AUX_OFF + par * GEMM6_AUX_BYTES;
        // the lane-derived addresses of the epilogue are recomputed per chunk (a handful of VALU) instead of living in registers across the
        // whole kernel: with two accumulator sets every loop-invariant the optimiser hoists is a spill (the opaque copy blocks the hoisting)
        int ln;
        VL2_LANE_ID_FRESH(ln);
        const int hi = ln >> 5, l31 = ln & 31;
        if constexpr (SWIGLU) {
            constexpr int mi = u / (NJ / 2), cbk = u % (NJ / 2);
            gemm6_epi_chunk<ACT, true, gh>(p, acc[S][mi][2 * cbk], acc[S][mi][2 * cbk + 1], m0 + wrow + mi * 32 + l31, wrow + mi * 32 + l31,
                                           wcol + cbk * 64, n0, hi, aux);
        } else {
            constexpr int mi = u / NJ, nj = u % NJ;
            gemm6_epi_chunk<ACT, false, gh>(p, acc[S][mi][nj], acc[S][mi][nj], m0 + wrow + mi * 32 + l31, wrow + mi * 32 + l31,
                                            wcol + nj * 32, n0, hi, aux);
        }
    };
    auto epi_all = [&](auto S_, int m0, int n0, int par) {
        vl2_static_for<0, NCH>([&](auto c_) { epi_chunk(S_, c_, m0, n0, par); });
    };

    int c_m0, c_n0, p_m0 = 0, p_n0 = 0, n_m0 = 0, n_n0 = 0;                 // origins of the current / previous / next tile
    unsigned nxt_v = 0;                                                     // dynamic form, wave 0 lane 0: the counter value drawn for the next tile
    const bool dyn0 = dyn && p.tile_first_dyn;                              // the first tile from the counter too: tickets 0, 1, 2, ... are the tiles
    const int t_base = dyn0 ? 0 : G;                                        // tile of ticket v = t_base + v
    int first_t0 = static_t0(0);
    if (dyn0) {
        // every workgroup draws its first tile: ~1.5-3 us at the head of the launch (256 tickets on one word), in exchange a workgroup that
        // is not resident when the grid starts (a CU short on some XCD) does not hold a statically assigned tile back by a whole tile time
        if (wave == 0 && lane == 0) VL2_LDS_I32(GEMM6_SLOT_OFF) = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        first_t0 = __builtin_amdgcn_readfirstlane(VL2_LDS_I32(GEMM6_SLOT_OFF));
        __syncthreads();
        // the first G tickets are drawn in (roughly) workgroup order, i.e. round-robin over the XCDs: keep gemm4's XCD-contiguous tile order
        if (first_t0 < G) first_t0 = xcd_remap(first_t0, G < ntiles ? G : ntiles);
    }
    if (dyn0 && first_t0 >= ntiles) {                                       // nothing left: count this workgroup as finished and leave
        if constexpr (!DB) if (wave == 0 && lane == 0) {
            const unsigned fin = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fin == (unsigned)G - 1u) {
                __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    tile_origin(first_t0, c_m0, c_n0);
    set_offsets(c_m0, c_n0);
    issue_aux(0, c_m0, c_n0);
    issue_slab(0, 0);
    issue_slab(1, 1);
    issue_slab(2, 2);
    wait_dma(2);                                                            // slab 0 (and the aux block) landed
    VL2_PHASE_BARRIER();
    if (grp == 1) VL2_PHASE_BARRIER();

    unsigned gc = 0;                                                        // slabs consumed so far (ring stage = gc & 3)
    auto load_frags = [&]() {
        const unsigned st = (gc & 3u) * GEMM4_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned ab = a_rd[ks] + st, bb = b_rd[ks] + st;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[ks][i] = *(const bf16x8*)(vl2_smem + ab + i * 2048);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[ks][j] = *(const bf16x8*)(vl2_smem + bb + j * 2048);
        }
    };
    auto mfmas = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[S][i][j] = VL2_MFMA32(fb[ks][j], fa[ks][i], acc[S][i][j]);
    };

    auto run_tile = [&](auto S_, int ord) -> bool {
        constexpr int S = decltype(S_)::value;
        const bool have_prev = ord > 0;
        bool have_next = false;
        // ---- peeled head phases: slab t + 3 of this tile; DB: one chunk of the previous tile's epilogue behind each phase's MFMAs
        vl2_static_for<0, NPEEL>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            if constexpr (!DB && t == 0) {                                  // group 0's epilogue of the previous tile: same barrier interval as group 1's
                if (grp == 0 && have_prev) epi_all(vl2_ic<0>{}, p_m0, p_n0, (ord - 1) & 1);
            }
            if constexpr (t == 0 && !DB) {
                // draw the next tile: ONE returning atomic per tile (lane 0 of wave 0), issued ahead of this phase's LDS-DMA.  Inline asm: a
                // compiler-tracked load result would make the waitcnt pass drain vmcnt at its first use (loads and stores pending together
                // count as out-of-order), i.e. empty the ring once per tile; here the ring's own counted waits retire it -- it is older than
                // the eight pieces that may still be in flight behind the wait of the NEXT phase, where the value is parked in LDS.
                if (dyn && wave == 0 && lane == 0) VL2_ATOMIC_INC_ASYNC(nxt_v, ctr);
            }
            issue_slab(gc + 3, t + 3);
            load_frags();
            wait_dma(2);
            VL2_WAIT_LGKMCNT0();
            VL2_PHASE_BARRIER();
            if constexpr (t == 1 && !DB) {
                if (dyn && wave == 0 && lane == 0) VL2_LDS_I32(GEMM6_SLOT_OFF) = t_base + (int)nxt_v;
            }
            if constexpr (t == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[S][i][j][r] = 0.f;
            }
            mfmas(S_);
            // DB: the MFMA group of a phase finishes its issue ~30 % before the LOAD group reaches the barrier (the matrix pipe is ~70 % busy
            // in steady state), so the ~75 vector instructions of a chunk fit into that slack and into the shadow of the last MFMAs; placed
            // BEHIND the MFMAs (not woven between them) the chunk's registers replace the fragments', which are dead once their MFMAs have
            // issued -- woven in, with two accumulator sets live, the allocator spills.
            if constexpr (DB) {
                __builtin_amdgcn_sched_barrier(0);
                if (have_prev) epi_chunk(vl2_ic<1 - S>{}, t_, p_m0, p_n0, (ord - 1) & 1);
            }
            VL2_PHASE_BARRIER();
            ++gc;
        });
        // ---- hot loop: gemm4's body
        for (int t = NPEEL; t < nt - 3; ++t) {
            issue_slab(gc + 3, t + 3);
            load_frags();
            wait_dma(2);
            VL2_WAIT_LGKMCNT0();
            VL2_PHASE_BARRIER();
            mfmas(S_);
            VL2_PHASE_BARRIER();
            ++gc;
        }
        // ---- tail: the next tile's first three slabs (or nothing)
        vl2_static_for<0, 3>([&](auto k_) {
            constexpr int k = decltype(k_)::value;                          // phase nt - 3 + k issues slab k of the next tile
            if constexpr (k == 0) {
                int n_t0;
                if (dyn) n_t0 = __builtin_amdgcn_readfirstlane(VL2_LDS_I32(GEMM6_SLOT_OFF));
                else n_t0 = static_t0(ord + 1);
                have_next = n_t0 < ntiles;
                if (have_next) {
                    tile_origin(n_t0, n_m0, n_n0);
                    set_offsets(n_m0, n_n0);
                    issue_aux((ord + 1) & 1, n_m0, n_n0);
                }
            }
            if (have_next) issue_slab(gc + 3, k);
            load_frags();
            if (have_next) wait_dma(2); else wait_dma(1 - k);
            VL2_WAIT_LGKMCNT0();
            VL2_PHASE_BARRIER();
            mfmas(S_);
            if constexpr (!DB && k == 2) {                                  // group 1's epilogue right behind its last MFMA phase
                if (grp == 1 && have_next) epi_all(vl2_ic<0>{}, c_m0, c_n0, ord & 1);
            }
            VL2_PHASE_BARRIER();
            ++gc;
        });
        p_m0 = c_m0;
        p_n0 = c_n0;
        c_m0 = n_m0;
        c_n0 = n_n0;
        return have_next;
    };
    int ord = 0;
    for (;;) {
        if (!run_tile(vl2_ic<0>{}, ord)) break;
        ++ord;
        if constexpr (DB) {
            if (!run_tile(vl2_ic<1>{}, ord)) break;
            ++ord;
        }
    }
    if (grp == 0) VL2_PHASE_BARRIER();
    // the last tile's epilogue (nothing left to hide it behind); `ord` = the ordinal of the last tile
    if constexpr (DB) {
        if (ord & 1) epi_all(vl2_ic<1>{}, p_m0, p_n0, 1);
        else epi_all(vl2_ic<0>{}, p_m0, p_n0, 0);
    } else {
        epi_all(vl2_ic<0>{}, p_m0, p_n0, ord & 1);
        // dynamic form: this workgroup has drawn its last (failing) ticket; the LAST workgroup to get here re-arms the block for the
        // next launch (everyone else has finished with the tile counter by then; launches that share a block are stream-ordered)
        if (dyn && wave == 0 && lane == 0) {
            const unsigned fin = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fin == (unsigned)G - 1u) {
                __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int ACT, bool SWIGLU, int BM, bool DB>
__global__ __launch_bounds__(512, 2) void gemm6_bf16_kernel(GemmArgs p) {
    gemm6_body<ACT, SWIGLU, BM, DB>(p);
}

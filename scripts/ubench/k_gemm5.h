// gemm5: one wave per SIMD, 128 x 128 output per wave.
//
// Why (profiles/r03_experiments.md): the 256x256 ping-pong kernel (gemm4, k_gemm.h) keeps the matrix pipe ~72 % busy and the
// board answers by clocking down to ~1.65 GHz -- the chip is power-limited on this workload, so what pays is FEWER non-MFMA
// events per FLOP and a fuller pipe at a lower clock, not more overlap.  gemm4 spends two waves per SIMD (wave tile 64 x 128:
// 12 fragment reads per 16 MFMAs) and two raw barriers per 32-deep slab to keep them in anti-phase.  Here a workgroup is FOUR
// waves, one per SIMD, each owning the whole 512-entry register file: a 128 x 128 accumulator block (4 x 4 MFMA 32x32x16 tiles
// = 256 accumulator registers) and two sets of 4 + 4 operand fragments.  Per 16-deep k-step a wave issues 16 MFMAs (512 cycles
// of its SIMD's matrix pipe) and only 8 ds_read_b128 (0.5 per MFMA instead of 0.75) + 4-5 LDS-DMA pieces; everything that is
// not an MFMA fits in the issue slots between MFMAs of the SAME wave (MI355X_MICROARCH.md: <= 5 single-issue instructions hide
// per 32x32x16 MFMA), so there is no partner wave to arbitrate with and ONE raw barrier per slab.
//
// Workgroup shapes (template WM x WN waves, WM * WN = 4): 2 x 2 = 256 x 256 tile (least operand traffic per FLOP),
// 1 x 4 = 128 x 512 (M granularity 128: the M = 1621 prefill), 4 x 1 = 512 x 128 (N granularity 128).
//
// K pipeline: slabs of 32 through a 4-stage LDS ring filled by LDS-DMA (`buffer_load_dwordx4 ... lds`), same LDS image and
// bank swizzle as gemm4 (gemm4_lds_off: 64-B rows, 4 rows per 256-B bank row, chunk c of row r at c ^ ((r >> 2) & 3); the
// swizzle sits on the per-lane SOURCE address and on the ds_read address).  Schedule of slab t (two k-steps ks = 0, 1):
//     A:  ds_read fragments (t, ks 1)          | 16 MFMAs on fragments (t, ks 0)
//         s_waitcnt vmcnt(PPW): slab t+1 landed (the PPW pieces of slab t+2 stay in flight);  s_barrier
//     B:  LDS-DMA slab t+3 -> stage of slab t-1 (every wave is past its reads of t-1: they were consumed before its MFMAs
//         (t-1, ks 1), which precede this barrier in program order);  ds_read fragments (t+1, ks 0) | 16 MFMAs (t, ks 1)
// so every ds_read has a whole 16-MFMA block (512 cycles) to return and every DMA ~2.5 slabs (2500 cycles) to land.
// Accumulation order over K is the same sequence of 16-deep MFMA steps as in every other GEMM kernel of this library and the
// epilogue is the shared gemm_store_patch -> a row's bits do not depend on which kernel computed it.
#pragma once
#include "k_gemm.h"

#define GEMM5_BK 32
#define GEMM5_STAGES 4

template <int WM, int WN>
struct Gemm5Geo {
    static constexpr int BM = WM * 128, BN = WN * 128;
    static constexpr int STAGE = (BM + BN) * 64;                  // bytes of one K-slab (A rows, then W rows)
    static constexpr int LDS = GEMM5_STAGES * STAGE;              // 128 KiB (2 x 2) / 160 KiB (1 x 4, 4 x 1)
    static constexpr int NA = BM / 64, NW = BN / 64;              // LDS-DMA pieces (1 KiB = 16 rows) per wave and slab
    static constexpr int PPW = NA + NW;
};

template <int ACT, bool SWIGLU, bool OUT_F32, int WM, int WN>
__device__ __forceinline__ void gemm5_body(const GemmArgs& p, int bid, int nwg) {
    using G = Gemm5Geo<WM, WN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware raster: every XCD walks a contiguous run of tiles, GR tile-rows deep (the A panel of a group stays in its L2)
    constexpr int GR = (WM == 1) ? 8 : (WM == 2 ? 4 : 2);          // ~1024 rows of A per raster group
    const int t0 = xcd_remap(bid, nwg);
    const int grp_sz = GR * p.tiles_n;
    const int first_m = (t0 / grp_sz) * GR;
    const int gm = (p.tiles_m - first_m) < GR ? (p.tiles_m - first_m) : GR;
    const int tm = first_m + (t0 % grp_sz) % gm, tn = (t0 % grp_sz) / gm;
    const int m0 = tm * G::BM, n0 = tn * G::BN;
    f32x2 rst[(G::BM + 255) / 256];
#pragma unroll
    for (int h = 0; h < (G::BM + 255) / 256; ++h) rst[h] = gemm_row_stats(p, m0, tid + 256 * h, G::BM);

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    // piece q of an operand slab = rows [16 q, 16 q + 16); this wave issues pieces q = 4 i + wave
    unsigned a_vo[G::NA], w_vo;
#pragma unroll
    for (int i = 0; i < G::NA; ++i) {
        const int slot = ((i * 4 + wave) << 6) + lane;
        const int R = slot >> 4, sp = slot & 15;
        const int row = 4 * R + (sp >> 2), chk = (sp & 3) ^ (R & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        a_vo[i] = ((unsigned)am * (unsigned)p.lda + chk * 8) * 2;
    }
    {
        const int slot = (wave << 6) + lane;
        const int R = slot >> 4, sp = slot & 15;
        const int row = 4 * R + (sp >> 2), chk = (sp & 3) ^ (R & 3);
        w_vo = ((unsigned)(n0 + row) * (unsigned)p.ldw + chk * 8) * 2;
    }
    const unsigned w_step = 128u * (unsigned)p.ldw;                // bytes between this wave's W pieces (64 rows)
    auto issue_dma = [&](int t) {
        const unsigned st = (unsigned)(t & (GEMM5_STAGES - 1)) * G::STAGE, kb = (unsigned)t * (GEMM5_BK * 2);
#pragma unroll
        for (int i = 0; i < G::NA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((i * 4 + wave) << 10)),
                                                     16, a_vo[i], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < G::NW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + G::BM * 64 + ((i * 4 + wave) << 10)),
                                                     16, w_vo, kb + i * w_step, 0, 0);
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb[2][4];                                     // [k-step][32-row block]

    const int nt = p.K / GEMM5_BK;
    const int frow = lane & 31, fchk = lane >> 5;
    unsigned a_rd[2], b_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_rd[ks] = gemm4_lds_off(wm * 128 + frow, ks * 2 + fchk);
        b_rd[ks] = G::BM * 64 + gemm4_lds_off(wn * 128 + frow, ks * 2 + fchk);
    }
#define GEMM5_READ(ks, st)                                                                                     \
    do {                                                                                                       \
        const unsigned ab_ = a_rd[ks] + (st), bb_ = b_rd[ks] + (st);                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[ks][i] = *(const bf16x8*)(vl2_smem + ab_ + i * 2048); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[ks][j] = *(const bf16x8*)(vl2_smem + bb_ + j * 2048); \
    } while (0)
#define GEMM5_MFMA(ks)                                                                                         \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0); \
    } while (0)
    // issue order inside a 16-MFMA block (sched_group_barrier pipelines, one scheduling region per block): the block OPENS with
    // an MFMA on fragments read a whole block earlier (so the wait in front of it covers no read of this block), then one memory
    // instruction per MFMA until the block's reads / LDS-DMA pieces are out, then the remaining MFMAs back to back.
#define GEMM5_MIX_A()                                                                                          \
    do {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                     \
    } while (0)
#define GEMM5_MIX_B(NDMA)                                                                                      \
    do {                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            if (q < (NDMA) - 8) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                             \
            else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                     \
    } while (0)

    issue_dma(0);
    if (nt > 1) issue_dma(1);
    if (nt > 2) issue_dma(2);
    if (nt > 2) VL2_WAIT_VMCNT(2 * G::PPW); else if (nt > 1) VL2_WAIT_VMCNT(G::PPW); else VL2_WAIT_VMCNT(0);
    VL2_PHASE_BARRIER();
    GEMM5_READ(0, 0u);

    int t = 0;
    for (; t + 3 < nt; ++t) {                                      // steady state: slabs t+1 .. t+3 exist
        const unsigned st = (unsigned)(t & 3) * G::STAGE, stn = (unsigned)((t + 1) & 3) * G::STAGE;
        GEMM5_READ(1, st);
        GEMM5_MFMA(0);
        GEMM5_MIX_A();
        VL2_WAIT_VMCNT(G::PPW);
        VL2_PHASE_BARRIER();
        issue_dma(t + 3);
        GEMM5_READ(0, stn);
        GEMM5_MFMA(1);
        GEMM5_MIX_B(G::PPW);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; t < nt; ++t) {                                          // drain: the last three slabs
        const unsigned st = (unsigned)(t & 3) * G::STAGE, stn = (unsigned)((t + 1) & 3) * G::STAGE;
        GEMM5_READ(1, st);
        GEMM5_MFMA(0);
        if (t + 3 <= nt) VL2_WAIT_VMCNT(G::PPW); else VL2_WAIT_VMCNT(0);
        VL2_PHASE_BARRIER();
        if (t + 1 < nt) GEMM5_READ(0, stn);
        GEMM5_MFMA(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    VL2_WAIT_LGKMCNT0();
    VL2_PHASE_BARRIER();                                           // every wave is done with the ring: it becomes epilogue space

    // ---- epilogue: the shared row-contiguous store path (gemm_store_patch), one wave-private 32 x 64 fp32 patch at a time.
    // The patches are wave-private and LDS operations of one wave execute in order, so no workgroup barrier is needed between
    // the passes: every wave streams its 128 x 128 block out at its own pace.
    float* ep = (float*)vl2_smem + wave * (32 * 68);
    float* rowtab = (float*)vl2_smem + 4 * (32 * 68);
#pragma unroll
    for (int h = 0; h < (G::BM + 255) / 256; ++h) gemm_park_row_stats(p, rowtab, rst[h], tid + 256 * h, G::BM);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    ep[row * 68 + ni * 32 + (lane & 31)] = acc[mi][nh * 2 + ni][r];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gemm_store_patch<ACT, SWIGLU, OUT_F32>(p, ep, m0 + wm * 128 + mi * 32, n0 + wn * 128 + nh * 64, lane, rowtab, wm * 128 + mi * 32);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#undef GEMM5_READ
#undef GEMM5_MFMA
#undef GEMM5_MIX_A
#undef GEMM5_MIX_B
}

template <int ACT, bool SWIGLU, bool OUT_F32, int WM, int WN>
__global__ __launch_bounds__(256, 1) void gemm5_bf16_kernel(GemmArgs p) {
    gemm5_body<ACT, SWIGLU, OUT_F32, WM, WN>(p, blockIdx.x, gridDim.x);
}

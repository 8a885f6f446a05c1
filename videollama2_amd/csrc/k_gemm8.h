// gemm8: 256 x 256 block tile on FOUR waves -- one wave per SIMD, each owning a 128 x 128 wave tile (4 x 4 accumulator blocks of 32 x 32 = 256
// accumulator registers of the 512 a lone wave may hold).  Round 5: the vendor's GEMMs (hipBLASLt, Tensile macro-tiles 256x256x64 / 256x192x64 with
// 256-thread workgroups and 128 x 128 / 128 x 96 wave tiles) measure 6-25 % faster than this library's two-waves-per-SIMD ping-pong kernels on every
// multi-round shape of the step (profiles/r05_vendor_gemm_ref_runH.txt).  What the geometry buys: a 128 x 128 wave tile reads 8 fragments for 16 MFMAs
// (0.5 per MFMA; the 64 x 128 tile of gemm4: 0.75, the 64 x 64 tiles: 1.0), and a lone wave per SIMD needs no phase barriers against a partner -- its
// own MFMAs (32 cycles each on the matrix pipe) leave ~5 issue slots per gap for the fragment reads and LDS-DMA of the NEXT k-step.
//
// Same slabs as gemm4 (K in 32-deep slabs through a 4-stage LDS ring by LDS-DMA, the same lane-linear swizzled image `gemm4_lds_off`), the same
// 32 x 32 x 16 MFMA in the same k order -> the same bits per element as every other kernel of the family (hash-checked; tests/test_gpu_ops.py).
// Pipeline per slab t (two k-steps):   MFMA(t, ks0) || ds_read fragments (t, ks1)
//                                      wait: slab t+1 landed (counted vmcnt) ; s_barrier ; LDS-DMA of slab t+3 into the slot of slab t-1
//                                      MFMA(t, ks1) || ds_read fragments (t+1, ks0)
// ONE barrier per slab (32 MFMAs = 1024 matrix-pipe cycles): at it every wave has finished reading slab t-1 (its fragments were in registers before
// its MFMAs issued) and has seen its own pieces of slab t+1 land.
//
// MEASURED (round 5, profiles/r05_experiments.md section 7; variant 9, never the automatic choice): bit-identical to the family; 8192^3 832.6 us against
// 850.0 for gemm4 and 677.6 for the vendor kernel on the same box (8192 x 4096 x 4096: 217.2 / 218.2 / 176.7) -- within 2 % of the ping-pong kernel, not
// the vendor's 20 %.  What holds it: an LDS-DMA piece costs the lone wave of a SIMD 60-100 cycles of issue (no partner wave to issue it under the MFMAs),
// eight pieces per 32-MFMA slab.  The vendor kernels stage through registers (plain loads + ds_write_b128, each of which fits an MFMA gap); that form was
// built here too: with ONE register set (one slab of global-load latency) it stalls on the loads (925 us), and with the two or three sets that cover the
// latency hipcc's allocator shuffles accumulators between the register files inside the loop (hundreds of v_accvgpr moves and scratch traffic per
// iteration; three formulations tried) -- that pipeline needs hand-written assembly, which this library does not carry.
#pragma once
#include "k_gemm.h"

#define GEMM8_LDS_BYTES GEMM4_LDS_BYTES

template <int ACT, bool SWIGLU, bool OUT_F32, bool TR = false, int EF = -1>
__device__ __forceinline__ void gemm8_body(const GemmArgs& p, int bid, int nwg) {
    static_assert(!(TR && OUT_F32), "gemm_store_tr writes bf16");
    constexpr int BM = 256, MI = 4, NJ = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = (wave >> 1) * 128, wcol = (wave & 1) * 128;

    const int t0 = xcd_remap(bid, nwg);
    const int grp_sz = 4 * p.tiles_n;                              // 4 tile-rows (1024 rows of A) per raster group, as gemm4
    const int first_m = (t0 / grp_sz) * 4;
    const int gm = (p.tiles_m - first_m) < 4 ? (p.tiles_m - first_m) : 4;
    const int tm = first_m + (t0 % grp_sz) % gm, tn = (t0 % grp_sz) / gm;
    const int m0 = tm * BM, n0 = tn * GEMM4_BN;
    f32x2 rst = {0.f, 1.f}, rowst[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rowst[i] = f32x2{0.f, 1.f};

    // this wave's LDS-DMA pieces of a slab: A pieces q = wave + 4 i (16 rows x 64 B each), W pieces likewise, i = 0..3 -- gemm4's slot map
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    unsigned a_vo[4], w_vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = ((wave + 4 * i) << 6) + lane;
        const int R = slot >> 4, sp = slot & 15;
        const int row = 4 * R + (sp >> 2), chk = (sp & 3) ^ (R & 3);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        a_vo[i] = ((unsigned)am * (unsigned)p.lda + chk * 8) * 2;
        w_vo[i] = ((unsigned)(n0 + row) * (unsigned)p.ldw + chk * 8) * 2;
    }
    auto issue_dma = [&](int t) {
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE, kb = (unsigned)t * (GEMM4_BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((wave + 4 * i) << 10)), 16, a_vo[i], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((wave + 4 * i) << 10)), 16, w_vo[i], kb, 0, 0);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][MI], fb[2][NJ];                   // [fragment set][tile]: set 0 = k-step 0, set 1 = k-step 1 of a slab

    const int nt = p.K / GEMM4_BK;
    const int frow = lane & 31, fchk = lane >> 5;
    unsigned a_rd[2], b_rd[2];                     // fragment read bases per k-step; tile i / j is +2048 B (32 rows)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_rd[ks] = gemm4_lds_off(wrow + frow, ks * 2 + fchk);
        b_rd[ks] = 16384 + gemm4_lds_off(wcol + frow, ks * 2 + fchk);
    }
    auto read_frags = [&](int set, unsigned st) {
        const unsigned ab = a_rd[set] + st, bb = b_rd[set] + st;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[set][i] = *(const bf16x8*)(vl2_smem + ab + i * 2048);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[set][j] = *(const bf16x8*)(vl2_smem + bb + j * 2048);
    };
    auto mfma16 = [&](int set) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = TR ? VL2_MFMA32(fb[set][j], fa[set][i], acc[i][j]) : VL2_MFMA32(fa[set][i], fb[set][j], acc[i][j]);
    };

    issue_dma(0);
    if (nt > 1) issue_dma(1);
    if (nt > 2) issue_dma(2);
    if constexpr (!TR) { rst = gemm_row_stats(p, m0, tid, BM); VL2_PIN2(rst[0], rst[1]); }
    else {
        gemm_tr_row_stats<MI>(p, m0 + wrow, lane, rowst);
#pragma unroll
        for (int i = 0; i < MI; ++i) VL2_PIN2(rowst[i][0], rowst[i][1]);
    }
    if (nt > 2) VL2_WAIT_VMCNT(16); else if (nt > 1) VL2_WAIT_VMCNT(8); else VL2_WAIT_VMCNT(0);
    VL2_PHASE_BARRIER();
    read_frags(0, 0u);

    // LDS-DMA of slab t+3 goes into the slot of slab t-1, which nobody reads any more once barrier t-1 is behind (the k-step-1 fragments of slab t-1
    // were in registers before it): HALF of the pieces (A) are issued in the first region of iteration t, the other half (W) behind barrier t, one piece
    // per three to four MFMAs -- a piece costs the lone wave ~60 cycles of issue (MI355X_MICROARCH.md), two MFMA slots, so bunching all eight behind the
    // barrier (hipcc's own placement) left the matrix pipe idle for ~350 cycles per slab.
    auto issue_half = [&](int tt, int half) {
        const unsigned st = (unsigned)(tt & 3) * GEMM4_STAGE, kb = (unsigned)tt * (GEMM4_BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (half == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((wave + 4 * i) << 10)), 16, a_vo[i], kb, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((wave + 4 * i) << 10)), 16, w_vo[i], kb, 0, 0);
        }
    };
    int t = 0;
    for (; t + 3 < nt; ++t) {                                       // steady state: no branches in the body, so the whole slab is two scheduling regions
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE, stn = (unsigned)((t + 1) & 3) * GEMM4_STAGE;
        read_frags(1, st);                                          // k-step 1 of slab t: lands under the MFMAs of k-step 0
        issue_half(t + 3, 0);
        mfma16(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
        for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);                          // (the wait stays BEHIND the 16 MFMAs: hipcc hoisted it to the second one)
        VL2_WAIT_VMCNT(12);                                         // slab t+1 has landed (slab t+2 and the first half of slab t+3 stay in flight)
        VL2_PHASE_BARRIER();                                        // ... for every wave; and every wave is past its reads of slab t-1
        issue_half(t + 3, 1);
        read_frags(0, stn);                                         // k-step 0 of slab t+1: lands under the MFMAs of k-step 1
        mfma16(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
#pragma unroll
        for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    for (; t < nt; ++t) {                                           // the last three slabs: nothing left to issue, counts run down
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE, stn = (unsigned)((t + 1) & 3) * GEMM4_STAGE;
        read_frags(1, st);
        mfma16(0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) VL2_WAIT_VMCNT(8); else VL2_WAIT_VMCNT(0);
        VL2_PHASE_BARRIER();
        if (t + 1 < nt) read_frags(0, stn);
        mfma16(1);
    }

    if constexpr (TR) {       // register-resident epilogue: the accumulators hold C^T, rows are lane-local
        gemm_store_tr<ACT, SWIGLU, MI, NJ, EF>(p, acc, m0 + wrow, n0 + wcol, lane, rowst);
        if constexpr (!SWIGLU) gemm_rows_ticket<256>(p, tm, m0, BM, tid);
        return;
    }
    VL2_PHASE_BARRIER();                                            // the ring is dead for every wave before the patches overwrite it
    float* ep = (float*)vl2_smem + wave * (32 * 68);
    float* rowtab = (float*)vl2_smem + 8 * (32 * 68);
    gemm_park_row_stats(p, rowtab, rst, tid, BM);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nh = 0; nh < NJ / 2; ++nh) {
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
    if constexpr (!SWIGLU && !OUT_F32) gemm_rows_ticket<256>(p, tm, m0, BM, tid);
}
template <int ACT, bool SWIGLU, bool OUT_F32, bool TR = false, int EF = -1>
__global__ __launch_bounds__(256, 1) void gemm8_bf16_kernel(GemmArgs p) {
    gemm8_body<ACT, SWIGLU, OUT_F32, TR, EF>(p, blockIdx.x, gridDim.x);
}

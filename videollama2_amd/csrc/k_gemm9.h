// gemm9: the 256 x 256 ping-pong GEMM of gemm4 on the OTHER bf16 matrix instruction, v_mfma_f32_16x16x32_bf16 (k_gemm.h gemm4_body is its twin on
// v_mfma_f32_32x32x16_bf16).  Round 5, found last: an MFMA-only loop on random operands sustains 2.12-2.15 PF with the 16 x 16 x 32 shape where the
// 32 x 32 x 16 shape sustains 1.79-1.88 (scripts/ubench/mfma_power16.hip) -- at the power limit this part runs at, the instruction with a quarter of the
// accumulator registers per issue is worth 15 %; the vendor's GEMM kernels use it.
//
// A DIFFERENT ARITHMETIC in the last bits: the 16 x 16 x 32 instruction sums 32 products per accumulation step where the family's instruction sums 16, so a
// dot product's association changes and this kernel is NOT bit-identical to the rest of the family (same fp32 accuracy: tests hold it to the fp32 oracle at the
// family's tolerance, and to itself across M -- a row's bits do not depend on how many rows the call has).  Therefore OPT-IN (VL2_GEMM_MFMA16 on a
// SwiGLU call = the decoder's gate/up projection, the step's dominant GEMM): the library's default keeps ONE association per dot product everywhere, which is
// what sharded == unsharded and batched == sequential pin.  Moving the WHOLE family is the next round's first item.
//
// MEASURED (round 5, profiles/r05_experiments.md section 11): against the same structure on 32 x 32 x 16 (variant 8), interleaved on one box -- 8192 x 4096 x 4096
// 217 -> 201 us (1266 -> 1367 TF/s; vendor 176 us), 8192^3 840 -> 801 (vendor 677), at 1.81 instead of 1.60 GHz for the same 1.39 kW; K = 1024 shapes (not at the
// power cap): equal; gate/up at S = 1621: 373 us against 341 for the default's mixed launch (6.3 tile rows: this kernel has no 128-row tail form) -> the stage
// switch loses 0.6 ms of prefill, which is why it is a switch.  rel-L2 against fp32: 1.66e-3 on every shape, the family's figure.
//
// Same slabs (K in 32-deep slabs = ONE k-step of the instruction), the same 4-stage LDS-DMA ring, the same two wave groups in anti-phase, the same wave tile
// (64 x 128 = 4 x 8 accumulator blocks of 16 x 16 = the same 128 accumulator registers).  What changes: fragment (i = row block of 16) = rows 16 i + (lane & 15),
// 16-B chunk lane >> 4 of the slab row -- ONE ds_read_b128, base + 1 KiB per block -- and the LDS swizzle that makes THAT access pattern conflict-free
// (`gemm9_lds_off`: chunk ^ H[(row >> 2) & 3], H = {0, 3, 2, 1}; checked against the ds_read_b128 lane groups of MI355X_MICROARCH.md: gemm4's swizzle is
// 2-way conflicted under this pattern); accumulators: register r of block (i, j) = C[16 i + 4 (lane >> 4) + r][16 j + (lane & 15)].
#pragma once
#include <type_traits>
#include "k_gemm.h"

__device__ __forceinline__ int gemm9_lds_off(int row, int chunk) {
    const int q = (row >> 2) & 3;
    return (((row >> 2) << 4) + ((row & 3) << 2) + (chunk ^ ((4 - q) & 3))) << 4;
}

// MODE: 0 = the shipped form (variant 16).  Lab forms, all the same bits, all measured on the device (profiles/r05_gemm_power_clock_ab.txt; 8192 x 4096 x 4096, MODE 0
// = 201.5 us): 1 (variant 17) the LDS-DMA of slab t+3 behind the load phase's fragment reads instead of ahead of them: 212 us; 2 (18) woven into the matrix phase, one
// piece per eight MFMAs (gemm4's WEAVE4): 201.3; 4 / 5 / 6 (20 / 21 / 22) the four pieces split 2 + 2 / 1 + 3 / 3 + 1 between load and matrix phase: 204.5-205.0 (and
// -3 ... -9 % on 8192^3); 3 (19) NO LDS-DMA: the slabs through REGISTERS (plain buffer loads two slabs ahead into two register sets, ds_write_b128 into the same LDS
// image one slab ahead -- the vendor kernels' path; 222 VGPRs, no spills, hipcc's own counted vmcnt(7 / 5 / 4)): 224 us.  7 / 8 (23 / 25) MODE 0 with s_memtime stamps
// (scripts/gemm9_phase_stamps.py): per wave and slab, LDS-DMA issue 252 cycles, fragment reads 250, the 32 MFMAs' issue 506, the two barriers ~50 + ~120; the undisturbed
// K loop 1225 cycles per slab against the matrix pipe's 1024.  So: the placement of the memory instructions is not what holds this structure.
// 9 (variant 26): 64-deep phases on a 5-stage ring in the full 160 KiB of LDS (half the barriers per FLOP; see the block below): +1.2 ... +2.7 % (8192^3 811 -> 793 us).
template <bool SWIGLU, int MODE = 0>
__device__ __forceinline__ void gemm9_body(const GemmArgs& p, int bid, int nwg) {
    constexpr int BM = 256, MI = 4, NJ = 8;                        // 16 x 16 accumulator blocks of a wave (64 rows x 128 columns)
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    unsigned long long c_k0 = 0, c_loop0 = 0, c_loop1 = 0;           // MODE 8 stamps: kernel entry | ring fill starts | K loop done (the fourth: epilogue done)
    if constexpr (MODE == 8) c_k0 = __builtin_readcyclecounter();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wrow = grp * 128 + (w4 >> 1) * 64, wcol = (w4 & 1) * 128;

    const int t0 = xcd_remap(bid, nwg);
    const int GD = p.tile_group > 0 ? p.tile_group : 4;              // row tiles that walk one W panel together on an XCD (lab variants 27-29: 8 / 2 / 6)
    const int grp_sz = GD * p.tiles_n;
    const int first_m = (t0 / grp_sz) * GD;
    const int gm = (p.tiles_m - first_m) < GD ? (p.tiles_m - first_m) : GD;
    const int tm = first_m + (t0 % grp_sz) % gm, tn = (t0 % grp_sz) / gm;
    const int m0 = tm * BM, n0 = tn * GEMM4_BN;
    f32x2 rst = {0.f, 1.f};

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    unsigned a_vo[2], w_vo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                   // gemm4's piece map with this kernel's swizzle on the SOURCE chunk
        const int aslot = ((grp * 8 + i * 4 + w4) << 6) + lane, wslot = grp * 512 + ((i * 4 + w4) << 6) + lane;
        const int Ra = aslot >> 4, spa = aslot & 15, Rw = wslot >> 4, spw = wslot & 15;
        const int rowa = 4 * Ra + (spa >> 2), chka = (spa & 3) ^ ((4 - (Ra & 3)) & 3);
        const int roww = 4 * Rw + (spw >> 2), chkw = (spw & 3) ^ ((4 - (Rw & 3)) & 3);
        int am = m0 + rowa;
        am = am < p.M ? am : p.M - 1;
        a_vo[i] = ((unsigned)am * (unsigned)p.lda + chka * 8) * 2;
        w_vo[i] = ((unsigned)(n0 + roww) * (unsigned)p.ldw + chkw * 8) * 2;
    }
    auto issue_dma = [&](int t) {
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE, kb = (unsigned)t * (GEMM4_BK * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * 8 + i * 4 + w4) << 10)), 16, a_vo[i], kb, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((grp * 8 + i * 4 + w4) << 10)), 16, w_vo[i], kb, 0, 0);
    };
    auto issue_piece = [&](int t, int tk, int i) {    // piece i (0, 1 = A; 2, 3 = W) of this wave's share of slab tk, into the ring slot of slab t
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE, kb = (unsigned)tk * (GEMM4_BK * 2);
        if (i < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((grp * 8 + i * 4 + w4) << 10)), 16, a_vo[i], kb, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((grp * 8 + (i - 2) * 4 + w4) << 10)), 16, w_vo[i - 2], kb, 0, 0);
    };
    auto wait_dma = [&](int slabs) {
        if (slabs >= 2) VL2_WAIT_VMCNT(8);
        else if (slabs == 1) VL2_WAIT_VMCNT(4);
        else VL2_WAIT_VMCNT(0);
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[MI], fb[NJ];

    const int nt = p.K / GEMM4_BK;
    const unsigned a_rd = gemm9_lds_off(wrow + (lane & 15), lane >> 4);                 // block i / j is + 1024 B (16 rows)
    const unsigned b_rd = 16384 + gemm9_lds_off(wcol + (lane & 15), lane >> 4);
    if constexpr (MODE == 9) {
        // 64-deep PHASES on a 5-stage ring (the whole 160 KiB): a load phase reads the fragments of TWO slabs (24 ds_read_b128 into 96 registers), a matrix phase
        // issues 64 MFMAs -- half the barriers per FLOP (the stamps of MODE 7: the two-barrier hand-over costs ~100 of every ~612 cycles).  Group 0 issues the WHOLE
        // slab 2T+3 in its LOAD(T) (eight pieces per wave), group 1 the whole slab 2T+4 in its LOAD(T); each wave sees its own pieces land (vmcnt(0)) at the end of
        // the matrix phase that follows, one barrier or more before anyone reads them: slab 2T+3 is read at LOAD(T+1), two phases after its issue; 2T+4 at LOAD(T+2).
        // Ring slot of slab s: s mod 5 -- the slot of pair T-1, whose last reader (group 1's LOAD(T-1)) is a barrier behind.
        const int np = nt >> 1;                                     // K % 64 == 0 (vl2_gemm): whole pairs
        bf16x8 fa2[2][MI], fb2[2][NJ];
        unsigned a_vo4[4], w_vo4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // this wave's eight pieces of a WHOLE slab: pieces w4 + 4 i of the 16 (A) + 16 (W)
            const int slot = ((i * 4 + w4) << 6) + lane;
            const int R = slot >> 4, sp = slot & 15;
            const int row = 4 * R + (sp >> 2), chk = (sp & 3) ^ ((4 - (R & 3)) & 3);
            int am = m0 + row;
            am = am < p.M ? am : p.M - 1;
            a_vo4[i] = ((unsigned)am * (unsigned)p.lda + chk * 8) * 2;
            w_vo4[i] = ((unsigned)(n0 + row) * (unsigned)p.ldw + chk * 8) * 2;
        }
        auto issue_slab = [&](int slab, int slot5) {
            const unsigned st = (unsigned)slot5 * GEMM4_STAGE, kb = (unsigned)slab * (GEMM4_BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(vl2_smem + st + ((i * 4 + w4) << 10)), 16, a_vo4[i], kb, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(vl2_smem + st + 16384 + ((i * 4 + w4) << 10)), 16, w_vo4[i], kb, 0, 0);
        };
        auto mod5 = [](int x) { return x >= 5 ? x - 5 : x; };
        if (grp == 0) issue_slab(0, 0);
        else { issue_slab(1, 1); if (nt > 2) issue_slab(2, 2); }
        rst = gemm_row_stats(p, m0, tid, BM);
        VL2_PIN2(rst[0], rst[1]);
        VL2_WAIT_VMCNT(0);
        VL2_PHASE_BARRIER();
        if (grp == 1) VL2_PHASE_BARRIER();
        int sa = 0;                                                 // ring slot of slab 2T
        for (int T = 0; T < np; ++T) {
            // ---------------- LOAD(T)
            const int mine = 2 * T + 3 + grp;
            if (mine < nt) issue_slab(mine, mod5(sa + 3 + grp));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned st = (unsigned)mod5(sa + h) * GEMM4_STAGE;
#pragma unroll
                for (int i = 0; i < MI; ++i) fa2[h][i] = *(const bf16x8*)(vl2_smem + st + a_rd + i * 1024);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb2[h][j] = *(const bf16x8*)(vl2_smem + st + b_rd + j * 1024);
            }
            VL2_WAIT_LGKMCNT0();
            VL2_PHASE_BARRIER();
            // ---------------- MFMA(T): 64 MFMAs, k ascending (slab 2T, then 2T+1): the same sums as the 32-deep phases
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = VL2_MFMA16(fa2[h][i], fb2[h][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            VL2_WAIT_VMCNT(0);                                      // this wave's pieces of the slab it issued in LOAD(T) have landed
            VL2_PHASE_BARRIER();
            sa = mod5(sa + 2);
        }
    } else {
    if constexpr (MODE == 8) { __builtin_amdgcn_sched_barrier(0); c_loop0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }   // MODE 8 (lab, variant 25): MODE 0 between stamps
    if constexpr (MODE == 7) {
        // MODE 0 with s_memtime stamps around the parts of the two phases (lab, variant 23): per workgroup and wave, the sums over the K loop of
        // [LDS-DMA issue | fragment reads issued and returned + the counted wait | barrier behind the load phase | 32 MFMAs issued | barrier behind the matrix phase]
        // and the slab count, as six u64 at sk_ws[(workgroup * 8 + wave) * 6].  (The stamps wait for lgkmcnt(0) themselves: placed where nothing else is pending.)
        auto now = [&]() { __builtin_amdgcn_sched_barrier(0); const unsigned long long c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); return c; };
        unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
        issue_dma(0);
        if (nt > 1) issue_dma(1);
        if (nt > 2) issue_dma(2);
        rst = gemm_row_stats(p, m0, tid, BM);
        VL2_PIN2(rst[0], rst[1]);
        wait_dma(nt > 2 ? 2 : nt > 1 ? 1 : 0);
        VL2_PHASE_BARRIER();
        if (grp == 1) VL2_PHASE_BARRIER();
        for (int t = 0; t < nt; ++t) {
            const unsigned long long c0 = now();
            if (t + 3 < nt) issue_dma(t + 3);
            const unsigned long long c1 = now();
            const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(vl2_smem + st + a_rd + i * 1024);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *(const bf16x8*)(vl2_smem + st + b_rd + j * 1024);
            wait_dma(nt - 2 - t);
            VL2_WAIT_LGKMCNT0();
            const unsigned long long c2 = now();
            VL2_PHASE_BARRIER();
            const unsigned long long c3 = now();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = VL2_MFMA16(fa[i], fb[j], acc[i][j]);
            const unsigned long long c4 = now();
            VL2_PHASE_BARRIER();
            const unsigned long long c5 = now();
            acc_t[0] += c1 - c0; acc_t[1] += c2 - c1; acc_t[2] += c3 - c2; acc_t[3] += c4 - c3; acc_t[4] += c5 - c4;
        }
        if (lane == 0 && p.sk_ws) {
            unsigned long long* o = (unsigned long long*)p.sk_ws + (size_t)(bid * 8 + wave) * 6;
#pragma unroll
            for (int k = 0; k < 5; ++k) o[k] = acc_t[k];
            o[5] = (unsigned long long)nt;
        }
    } else if constexpr (MODE >= 4) {
        constexpr int NW = MODE == 4 ? 2 : MODE == 5 ? 3 : 1;        // pieces woven into the matrix phase (the last NW of the four)
        // branch-free: always three slabs in the prologue and one per iteration -- past the end of K the LAST slab again, into a dead slot; so the counted wait
        // is one number: behind the load phase's pieces, slab t+2 (4) and the load phase's share of slab t+3 (4 - NW) may be in flight
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(q, q < nt ? q : nt - 1, i);
        rst = gemm_row_stats(p, m0, tid, BM);
        VL2_PIN2(rst[0], rst[1]);
        VL2_WAIT_VMCNT(8);
        VL2_PHASE_BARRIER();
        if (grp == 1) VL2_PHASE_BARRIER();
        for (int t = 0; t < nt; ++t) {
            const int tk = t + 3 < nt ? t + 3 : nt - 1;
#pragma unroll
            for (int i = 0; i < 4 - NW; ++i) issue_piece(t + 3, tk, i);
            const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(vl2_smem + st + a_rd + i * 1024);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *(const bf16x8*)(vl2_smem + st + b_rd + j * 1024);
            VL2_WAIT_VMCNT(8 - NW);
            VL2_WAIT_LGKMCNT0();
            VL2_PHASE_BARRIER();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = VL2_MFMA16(fa[i], fb[j], acc[i][j]);
                    constexpr int STEP = 32 / NW;
                    const int idx = i * NJ + j;
                    if (idx % STEP == STEP / 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(t + 3, tk, 4 - NW + idx / STEP);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            VL2_PHASE_BARRIER();
        }
        VL2_WAIT_VMCNT(0);
    } else if constexpr (MODE == 3) {
        // register-staged pipeline.  At LOAD(t): W(t+1) = this wave's four pieces of slab t+1 from register set (t+1) & 1 into ring slot (t+1) & 3 (the same
        // lane-linear image the LDS-DMA writes: piece base + 16 lane, conflict-free), G(t+3) = the loads of slab t+3 into the set just written out, R(t) = the
        // fragment reads of slab t; every wave wrote its pieces of slab t at ITS LOAD(t-1), one barrier or more before anyone's R(t).  hipcc counts vmcnt itself.
        u32x4 ra[2][2], rw[2][2];
        const unsigned wr_a = (unsigned)((grp * 8 + w4) << 10) + lane * 16, wr_w = 16384u + wr_a;        // piece i is + 4 KiB
        auto G = [&](auto set, int t) {
            constexpr int S = decltype(set)::value;
            const unsigned kb = (unsigned)t * (GEMM4_BK * 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[S][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_vo[i], kb, 0));
                rw[S][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, w_vo[i], kb, 0));
            }
        };
        auto W = [&](auto set, int t) {
            constexpr int S = decltype(set)::value;
            const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *(u32x4*)(vl2_smem + st + wr_a + i * 4096) = ra[S][i];
                *(u32x4*)(vl2_smem + st + wr_w + i * 4096) = rw[S][i];
            }
        };
        auto slab = [&](auto set1, int t, auto full) {             // set1 = the register set of slab t+1 (and t+3); full: t + 3 < nt is known (no branches)
            if (decltype(full)::value || t + 1 < nt) W(set1, t + 1);
            if (decltype(full)::value || t + 3 < nt) G(set1, t + 3);
            const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(vl2_smem + st + a_rd + i * 1024);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *(const bf16x8*)(vl2_smem + st + b_rd + j * 1024);
            VL2_WAIT_LGKMCNT0();
            VL2_PHASE_BARRIER();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = VL2_MFMA16(fa[i], fb[j], acc[i][j]);
            VL2_PHASE_BARRIER();
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        G(S0{}, 0);
        if (nt > 1) G(S1{}, 1);
        rst = gemm_row_stats(p, m0, tid, BM);
        VL2_PIN2(rst[0], rst[1]);
        W(S0{}, 0);
        if (nt > 2) G(S0{}, 2);
        VL2_WAIT_LGKMCNT0();
        VL2_PHASE_BARRIER();
        if (grp == 1) VL2_PHASE_BARRIER();
        int t = 0;
        for (; t + 4 < nt; t += 2) {                                // steady state without branches: hipcc's vmcnt for W(t+1) then leaves the loads of slab t+2 in flight
            slab(S1{}, t, std::true_type{});                        // (with the tail's conditions inside the loop it merged the paths to vmcnt(0): one slab of latency cover)
            slab(S0{}, t + 1, std::true_type{});
        }
        for (; t + 1 < nt; t += 2) {                                // the last three or four slabs (t stays even)
            slab(S1{}, t, std::false_type{});
            slab(S0{}, t + 1, std::false_type{});
        }
        if (t < nt) slab(S1{}, t, std::false_type{});
    } else {
    issue_dma(0);
    if (nt > 1) issue_dma(1);
    if (nt > 2) issue_dma(2);
    rst = gemm_row_stats(p, m0, tid, BM);
    VL2_PIN2(rst[0], rst[1]);
    wait_dma(nt > 2 ? 2 : (nt > 1 && MODE != 2) ? 1 : 0);          // (woven form: its in-loop wait keeps ONE slab in flight, which at nt = 2 would be slab 1 itself)
    VL2_PHASE_BARRIER();

    if (grp == 1) VL2_PHASE_BARRIER();
    for (int t = 0; t < nt; ++t) {
        // ---------------- LOAD(t)
        if ((MODE == 0 || MODE == 8) && t + 3 < nt) issue_dma(t + 3);
        const unsigned st = (unsigned)(t & 3) * GEMM4_STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(vl2_smem + st + a_rd + i * 1024);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = *(const bf16x8*)(vl2_smem + st + b_rd + j * 1024);
        if constexpr (MODE == 1) {
            __builtin_amdgcn_sched_barrier(0);
            if (t + 3 < nt) issue_dma(t + 3);
        }
        if constexpr (MODE == 2) wait_dma(1); else wait_dma(nt - 2 - t);      // (woven: one newer slab in flight in every iteration, see gemm4_body)
        VL2_WAIT_LGKMCNT0();
        VL2_PHASE_BARRIER();
        // ---------------- MFMA(t): 32 x v_mfma_f32_16x16x32_bf16 = the slab's one k-step
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = VL2_MFMA16(fa[i], fb[j], acc[i][j]);
                if constexpr (MODE == 2) {
                    const int idx = i * NJ + j;
                    if (idx % 8 == 2) {                                      // behind MFMA 2, 10, 18, 26: one piece each (past the end of K: the last slab again, into the dead slot)
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(t + 3, t + 3 < nt ? t + 3 : nt - 1, idx / 8);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        VL2_PHASE_BARRIER();
    }
    if constexpr (MODE == 2) VL2_WAIT_VMCNT(0);
    }
    if constexpr (MODE == 8) { __builtin_amdgcn_sched_barrier(0); c_loop1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
    }
    if (grp == 0) VL2_PHASE_BARRIER();

    // ---- epilogue: 32 x 64 fp32 patches through the ring's memory, gemm_store_patch (bias / RMS row scale / SwiGLU / residual as the family)
    float* ep = (float*)vl2_smem + wave * (32 * 68);
    float* rowtab = (float*)vl2_smem + 8 * (32 * 68);
    gemm_park_row_stats(p, rowtab, rst, tid, BM);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ep[(ib * 16 + 4 * (lane >> 4) + r) * 68 + jb * 16 + (lane & 15)] = acc[mi * 2 + ib][nh * 4 + jb][r];
            __builtin_amdgcn_wave_barrier();
            gemm_store_patch<ACT_NONE, SWIGLU, false>(p, ep, m0 + wrow + mi * 32, n0 + wcol + nh * 64, lane, rowtab, wrow + mi * 32);
            __builtin_amdgcn_wave_barrier();
        }
    if constexpr (!SWIGLU && MODE == 0) gemm_rows_ticket<512>(p, tm, m0, BM, tid);       // producer-side finalize of the row statistics (o / down); (MODE 9 owns all 160 KiB of LDS: no room for the ticket's word)
    if constexpr (MODE == 8) {          // [ring fill + K loop | setup before it | epilogue (stores issued, not drained) | entry stamp | 0 | slabs]
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long c_end = __builtin_readcyclecounter();
        if (lane == 0 && p.sk_ws) {
            unsigned long long* o = (unsigned long long*)p.sk_ws + (size_t)(bid * 8 + wave) * 6;
            o[0] = c_loop1 - c_loop0; o[1] = c_loop0 - c_k0; o[2] = c_end - c_loop1; o[3] = c_k0; o[4] = c_end; o[5] = (unsigned long long)nt;
        }
    }
}
template <bool SWIGLU, int MODE = 0>
__global__ __launch_bounds__(512, 2) void gemm9_bf16_kernel(GemmArgs p) {
    gemm9_body<SWIGLU, MODE>(p, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------
// The 128 x 128 one-round tile (k_gemm.h gemm_l8_body: eight waves, wave tile 32 x 64, 4-stage 128 KiB LDS-DMA ring, one barrier per 64-deep
// K-tile) on v_mfma_f32_16x16x32_bf16: the TAIL tiles of a row-split call whose leading rows run on gemm9_body (gemm_mix16_bf16_kernel below).
// Same LDS image and ring as gemm_l8_body; a K-tile is two k-steps of the instruction (k ascending), fragment (block of 16 rows) = rows
// 16 b + (lane & 15), 16-B chunk 4 kk + (lane >> 4) -- the k index function of gemm9_body, so every dot product is the SAME sequence of
// 32-product accumulation steps: a row has gemm9's bits whichever of the two bodies computes it.  Accumulators: 2 x 4 blocks of 16 x 16
// (register r of block (i, j) = C[16 i + 4 (lane >> 4) + r][16 j + (lane & 15)]), through the family's LDS patch epilogue.
template <bool SWIGLU>
__device__ __forceinline__ void gemm_l8_16_body(const GemmArgs& p, int bid, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                    // wm 0..3 (32 rows each), wn 0..1 (64 columns each)
    const int t = xcd_remap(bid, nwg);
    const int tm = t % p.tiles_m, tn = t / p.tiles_m;            // consecutive workgroups share a W panel
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int nt = p.K / GEMM_BK;
    const f32x2 rst = gemm_row_stats(p, m0, tid, GEMM_BM);
    // a wave whose 32 rows all lie past M (the ragged last row tile: 85 live rows of gate/up's tail at S = 1621 leave wave row 3 empty) moves its share of
    // every slab and meets every barrier but issues no fragment read and no MFMA: nothing of it is stored, and on a part that runs these launches at its
    // power limit the matrix work it skips is clock for the others
    const bool live = m0 + wm * 32 < p.M || p.tile_group < 0;      // (tile_group < 0: lab variant 28, the skip switched off for A/B runs)

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
    unsigned a_vo[2], w_vo;                                     // 2 A pieces + 2 W pieces per wave (piece i = +64 rows): gemm_l8_body's piece map
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
    const int frow = lane & 15, fchk = lane >> 4;
    unsigned a_rd[2][2], b_rd[2][4];                            // [k-step][block]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a_rd[kk][i] = gemm_lds_off(wm * 32 + i * 16 + frow, kk * 4 + fchk);
#pragma unroll
        for (int j = 0; j < 4; ++j) b_rd[kk][j] = 16384 + gemm_lds_off(wn * 64 + j * 16 + frow, kk * 4 + fchk);
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
        if (!live) continue;                                        // (see `live` above: loads and barriers only)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(vl2_smem + lds_buf + a_rd[kk][i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(vl2_smem + lds_buf + b_rd[kk][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = VL2_MFMA16(fa[i], fb[j], acc[i][j]);
        }
    }
    VL2_WAIT_LGKMCNT0();
    VL2_PHASE_BARRIER();

    float* ep = (float*)vl2_smem + wave * (32 * 68);
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ep[(ib * 16 + 4 * (lane >> 4) + r) * 68 + jb * 16 + (lane & 15)] = acc[ib][jb][r];
    float* rowtab = (float*)vl2_smem + 8 * (32 * 68);
    gemm_park_row_stats(p, rowtab, rst, tid, GEMM_BM);
    __syncthreads();
    gemm_store_patch<ACT_NONE, SWIGLU, false>(p, ep, m0 + wm * 32, n0 + wn * 64, lane, rowtab, wm * 32);
    if constexpr (!SWIGLU) gemm_rows_ticket<512>(p, tm, m0, GEMM_BM, tid);
}
template <bool SWIGLU>
__global__ __launch_bounds__(512, 1) void gemm_l8_16_bf16_kernel(GemmArgs p) {
    gemm_l8_16_body<SWIGLU>(p, blockIdx.x, gridDim.x);
}

// ONE launch for a row-split call on the 16 x 16 x 32 instruction (the twin of k_gemm.h gemm_mix_bf16_kernel): workgroups [0, n_big) run gemm9_body on
// the leading whole 256-row tiles (`pb`), workgroups [n_big, grid) the 128 x 128 body above on the tail rows (`ps`).  Round 5 measured gemm9 alone on
// gate/up at S = 1621 at 373 us against 341 for the default's mixed launch -- seven 256-row tiles for 6.3 tile rows of work; this form keeps the mix.
// MODE = the big tiles' form: 0 = 32-deep phases on the 4-stage ring (128 KiB of LDS), 9 = 64-deep phases on the 5-stage ring (160 KiB; K % 64 == 0).
template <bool SWIGLU, int MODE = 0>
__global__ __launch_bounds__(512, 2) void gemm_mix16_bf16_kernel(GemmArgs pb, GemmArgs ps, int n_big) {
    if ((int)blockIdx.x < n_big) gemm9_body<SWIGLU, MODE>(pb, blockIdx.x, n_big);
    else gemm_l8_16_body<SWIGLU>(ps, (int)blockIdx.x - n_big, (int)gridDim.x - n_big);
}

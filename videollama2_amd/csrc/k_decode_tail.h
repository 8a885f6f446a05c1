// Decode "tail engine": o_proj -> (+residual) -> RMSNorm -> gate/up + SwiGLU -> down_proj (+residual) of ONE decoder layer and ONE
// token (HF:models/mistral/modeling_mistral.py:202-241 after the attention) as ONE persistent launch instead of three GEMV launches.
//
// Why (profiles/r03_experiments.md section 5): every launch of the decode graph costs ~4.5 us whatever it does, the weight
// stream itself runs at 6.9 TB/s, so 30 % of a decode step is fixed cost.  A grid barrier costs about what a launch boundary costs -- what an
// engine can win is (a) the weight stream does not stop at the boundary: every wave has the first loads of its NEXT phase's rows in flight
// (16-B nontemporal loads straight to registers, 8-16 KB per wave, 32-64 MB chip-wide = 5-9 us of stream) when it arrives at the barrier and
// consumes them behind it; (b) x is staged (and, for gate/up, RMS-normalised) once per CU and phase instead of once per 4-row workgroup
// (3584 workgroups of the gate/up GEMV each read x and a 16 KB vector of ones).
//
// Geometry: G workgroups (one per CU), 1024 threads = 16 waves; wave gw = 16 b + w owns rows gw, gw + 16 G, ... of every phase.
// Per-row arithmetic = gemv_bf16_kernel's (k_decode.h), lane for lane and in the same order -> the same bits as the three GEMV launches
// (graph == eager, batched row == single row stay bit-exact).
//
// Hand-off between phases (guide G16, form R1): a row's result is ONE bf16, stored write-through (agent-scope relaxed atomic store = `sc0 sc1`)
// and drained by the storing wave; raw s_barrier; the barrier wave (15, which holds no prefetch) arrives on an XCD-hierarchical counter
// (8 groups of G / 8 by blockIdx, then one top word; one generation word per group to poll), polls relaxed with s_sleep, does ONE agent-scope
// acquire (invalidates this CU's vector L1), raw s_barrier; then everyone stages the vector with plain loads.  Every spin is bounded: a
// timeout raises bar[24] and lets the launch finish with garbage instead of hanging the GPU.  The barrier words are zeroed by the argmax
// launch that opens every decode step (k_decode.h argmax_kernel `zero`), so an aborted step cannot poison the next one.
//
// PHASES: bit 0 = o_proj, 1 = gate/up, 2 = down.  7 = the engine (grid barriers between the phases).  A single-phase instantiation has no
// barrier: the CPU test build (which runs workgroups one after the other) launches 1, 2, 4 back to back through the same code.
#pragma once
#include "k_decode.h"

struct TailArgs {
    const bf16_t* Wo;  const bf16_t* Wgu;  const bf16_t* Wd;     // [D, QD], [2I, D] (gate/up blocks of 64 rows), [D, I]
    int ldwo, ldwgu, ldwd;
    const bf16_t* o;        // [QD] attention output of this token
    const bf16_t* x0;       // [D]  residual stream entering the layer
    bf16_t* x1;             // [D]  x0 + Wo o                          (phase 0 out, phase 1 in, phase 2 residual)
    bf16_t* act;            // [I]  silu(gate) * up of RMSNorm(x1)     (phase 1 out, phase 2 in)
    bf16_t* xout;           // [D]  x1 + Wd act                        (phase 2 out; may be x0)
    int D, QD, I;
    float eps;
    unsigned* bar;          // >= 32 zeroed words: [0..7] group arrivals, [8] top arrivals, [9] workgroups finished, [16..23] generation per group, [24] timeout
};

#define TAIL_WAVES 16
#define TAIL_BARW 15                      // the wave that runs the grid-barrier protocol (it prefetches nothing across a barrier)
#define TAIL_SPIN_LIMIT (1 << 20)
#define TAIL_MAX_ROWS 8                   // rows of one phase per wave (the launcher checks I <= 8 * 16 * workgroups)

#ifndef VL2_TAIL_STORE_BF16               // write-through store of one bf16 (the CPU test build defines its own)
#define VL2_TAIL_STORE_BF16(ptr, val) __hip_atomic_store((unsigned short*)(ptr), (unsigned short)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// grid barrier number `gen` (1, 2, ...) of this launch; called by ALL threads of every workgroup
__device__ __forceinline__ void tail_grid_barrier(unsigned* bar, unsigned gen, int nwg) {
    __builtin_amdgcn_s_barrier();                                     // every wave has drained its write-through stores before this
    if (threadIdx.x == TAIL_BARW * 64) {
        const int grp = blockIdx.x & 7, per = nwg >> 3;
        const unsigned old = __hip_atomic_fetch_add(bar + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (unsigned)per * gen) {
            const unsigned old2 = __hip_atomic_fetch_add(bar + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old2 + 1u == 8u * gen) {
#pragma unroll
                for (int x = 0; x < 8; ++x) __hip_atomic_store(bar + 16 + x, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        int spins = 0;
        while (__hip_atomic_load(bar + 16 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > TAIL_SPIN_LIMIT) { __hip_atomic_store(bar + 24, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // this CU's vector L1 holds nothing stale of the vector staged next
    }
    __builtin_amdgcn_s_barrier();
}

template <int PHASES>
__global__ __launch_bounds__(1024) void decode_tail_kernel(TailArgs p) {
#pragma clang fp reassociate(off)                                     // the RMSNorm arithmetic in gemv_bf16_kernel's order (k_decode.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    __shared__ float red[4];
    __shared__ float outs[TAIL_WAVES][TAIL_MAX_ROWS];                // a wave's results of the current phase (stored behind the row loop)
    bf16_t* xs = (bf16_t*)vl2_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = (int)gridDim.x, gw = (int)blockIdx.x * TAIL_WAVES + wave, nwaves = nwg * TAIL_WAVES;
    constexpr bool ENGINE = PHASES == 7;
    u32x4 wv[8], uv[8];

    // 8 (x 2 with SWIGLU) nontemporal 16-B loads of output row j, vectors [v0 + 64 i + lane], i = 0..7
    auto issue_row = [&](const bf16_t* W, int ldw, int nvec, bool swiglu, int j, int v0) {
        const int row0 = swiglu ? (j >> 5) * 64 + (j & 31) : j;
        const bf16_t* w0p = W + (size_t)row0 * ldw;
        const bf16_t* w1p = w0p + (size_t)32 * ldw;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = v0 + i * 64 + lane;
            if (v < nvec) {
                wv[i] = __builtin_nontemporal_load((const u32x4*)(w0p + (size_t)v * 8));
                if (swiglu) uv[i] = __builtin_nontemporal_load((const u32x4*)(w1p + (size_t)v * 8));
            }
        }
    };
    // x [K] bf16 -> LDS, optionally RMS-normalised (HF MistralRMSNorm with the weight folded into the projection: fp32 statistics in the
    // summation tree of gemv_bf16_kernel -- 256 threads x (k, k + 2048), wave sums, ((w0 + w1) + (w2 + w3)) -- result rounded to bf16)
    auto stage_x = [&](const bf16_t* x, int K, bool norm) {
        float rstd = 1.f;
        if (norm) {
            if (tid < 256) {
                float ss = 0.f;
                for (int k = tid * 8; k < K; k += 2048) {
                    float v[8];
                    unpack8(*(const u32x4*)(x + k), v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(v[j], v[j], ss);
                }
                ss = wave_sum(ss);
                if (lane == 0) red[wave] = ss;
            }
            __syncthreads();
            rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + p.eps);
        }
        for (int k = tid * 8; k < K; k += 8192) {
            u32x4 raw = *(const u32x4*)(x + k);
            if (norm) {
                float v[8];
                unpack8(raw, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (v[j] * rstd) * 1.0f;       // (the folded norm weight is 1: the product is exact)
                raw = pack8(v);
            }
            *(u32x4*)(xs + k) = raw;
        }
        __syncthreads();
    };
    // dot products of the loaded vectors of one pass with x in LDS
    auto dots = [&](int nvec, bool swiglu, int v0, float& a0, float& a1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = v0 + i * 64 + lane;
            if (v < nvec) {
                const u32x4 xv = *(const u32x4*)(xs + (size_t)v * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a0 = dot2_bf16(wv[i][q], xv[q], a0);
                    if (swiglu) a1 = dot2_bf16(uv[i][q], xv[q], a1);
                }
            }
        }
    };
    auto store_out = [&](bf16_t* dst, float o) {
        if (ENGINE) VL2_TAIL_STORE_BF16(dst, f2bf(o));               // write-through: read by other CUs behind the grid barrier
        else *dst = f2bf(o);
    };
    unsigned gen = 0;
    bool pre = false;                                                 // the first pass of the next phase's first row is already in flight

    // One phase: rows gw, gw + nwaves, ... < N of W [N, K] against x (already being staged by the caller's stage_x).  The FIRST pass of the
    // first row is in flight when this is called (issued at the phase start, or before the grid barrier) and is consumed first, on its own
    // path: hipcc's waitcnt pass then sees no load pending when the loop below re-issues into the same registers (with the first row inside
    // the loop it drains vmcnt before EVERY load it issues -- eight serialized memory round trips per row, measured 3.77 ms per token).
    // fin(row ordinal r, row j, a0, a1) -> the row's value, parked in LDS; the stores leave behind the loop (a pending store next to the row
    // loads has the same effect on the pass).
    auto run_rows = [&](const bf16_t* W, int ldw, int K, bool swiglu, int N, auto fin) {
        const int nvec = K >> 3;
        if (gw < N) {
            float a0 = 0.f, a1 = 0.f;
            dots(nvec, swiglu, 0, a0, a1);
            for (int v0 = 512; v0 < nvec; v0 += 512) {
                issue_row(W, ldw, nvec, swiglu, gw, v0);
                dots(nvec, swiglu, v0, a0, a1);
            }
            a0 = wave_sum(a0);
            if (swiglu) a1 = wave_sum(a1);
            if (lane == 0) outs[wave][0] = fin(gw, a0, a1);
        }
        int r = 1;
        for (int j = gw + nwaves; j < N; j += nwaves, ++r) {
            float a0 = 0.f, a1 = 0.f;
            for (int v0 = 0; v0 < nvec; v0 += 512) {
                issue_row(W, ldw, nvec, swiglu, j, v0);
                dots(nvec, swiglu, v0, a0, a1);
            }
            a0 = wave_sum(a0);
            if (swiglu) a1 = wave_sum(a1);
            if (lane == 0) outs[wave][r] = fin(j, a0, a1);
        }
    };
    // ---------------- phase 0: x1 = x0 + Wo o
    if constexpr (PHASES & 1) {
        if (gw < p.D) issue_row(p.Wo, p.ldwo, p.QD >> 3, false, gw, 0);
        stage_x(p.o, p.QD, false);
        run_rows(p.Wo, p.ldwo, p.QD, false, p.D, [&](int j, float a0, float) { return a0 + bf2f(p.x0[j]); });
        if (lane == 0)
            for (int j = gw, r = 0; j < p.D; j += nwaves, ++r) store_out(p.x1 + j, outs[wave][r]);
    }
    if constexpr (ENGINE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's write-through stores have left
        if (wave != TAIL_BARW && gw < p.I) { issue_row(p.Wgu, p.ldwgu, p.D >> 3, true, gw, 0); pre = true; }
        tail_grid_barrier(p.bar, ++gen, nwg);
    }
    // ---------------- phase 1: act = silu(gate) * up of RMSNorm(x1)
    if constexpr (PHASES & 2) {
        if (!pre && gw < p.I) issue_row(p.Wgu, p.ldwgu, p.D >> 3, true, gw, 0);
        stage_x(p.x1, p.D, true);
        run_rows(p.Wgu, p.ldwgu, p.D, true, p.I, [&](int, float a0, float a1) { return silu_f(a0) * a1; });
        if (lane == 0)
            for (int j = gw, r = 0; j < p.I; j += nwaves, ++r) store_out(p.act + j, outs[wave][r]);
        pre = false;
    }
    if constexpr (ENGINE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave != TAIL_BARW && gw < p.D) { issue_row(p.Wd, p.ldwd, p.I >> 3, false, gw, 0); pre = true; }
        tail_grid_barrier(p.bar, ++gen, nwg);
    }
    // ---------------- phase 2: xout = x1 + Wd act
    if constexpr (PHASES & 4) {
        if (!pre && gw < p.D) issue_row(p.Wd, p.ldwd, p.I >> 3, false, gw, 0);
        stage_x(p.act, p.I, false);
        run_rows(p.Wd, p.ldwd, p.I, false, p.D, [&](int j, float a0, float) { return a0 + bf2f(p.x1[j]); });
        if (lane == 0)
            for (int j = gw, r = 0; j < p.D; j += nwaves, ++r) p.xout[j] = f2bf(outs[wave][r]);
    }
    if constexpr (ENGINE) {
        // the last workgroup to finish re-arms the barrier words (everyone else has left every poll by then)
        __builtin_amdgcn_s_barrier();
        if (tid == TAIL_BARW * 64) {
            const unsigned fin = __hip_atomic_fetch_add(p.bar + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fin + 1u == (unsigned)nwg) {
#pragma unroll
                for (int x = 0; x < 10; ++x) __hip_atomic_store(p.bar + x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int x = 16; x < 24; ++x) __hip_atomic_store(p.bar + x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The four-launch decode layer (round 6): per token and layer
//   attn_oproj_kernel : RoPE + KV-cache append + the 64-key attention slices + the combine of their partials + the o_proj GEMV (+ residual)
//                       in ONE launch: the o_proj weight rows (they depend on nothing) stream into registers while the attention runs;
//                       two in-launch hand-offs (slices -> combine per kv head -> everyone).  Replaces {attn_decode_kernel,
//                       attn_decode_combine_kernel, o_proj gemv_bf16_kernel} = three launches.
//   gemv2_bf16_kernel : gemv_bf16_kernel with the activation vector requested BEFORE the weight rows (a wave's loads return in order:
//                       x queued behind eight 16-B weight loads per lane cannot be staged before they land).
// Every kernel gives the bits of the launches it replaces (HF:models/mistral/modeling_mistral.py MistralAttention at q_len 1 on a
// DynamicCache; videollama2/model/videollama2_mistral.py:110-144 generate -> HF GenerationMixin._sample, one token per forward).
#pragma once
#include "k_decode.h"

// ------------------------------------------------------------------------------------------------- attention + combine + o_proj, one launch
struct AttnOprojArgs {
    const bf16_t* qkv;      // [(nh + 2 nkv) * 128] the un-roped projection of the new token (previous launch)
    bf16_t* kcache;         // [nkv][smax][128]
    bf16_t* vcache;
    const float* cos_t;     // [maxpos][64]
    const float* sin_t;
    float* partial;         // [nh][nsplit_cap][130] {m, l, o[128]} of every slice (written write-through, read back inside the launch)
    bf16_t* o;              // [nh * 128] the attention output (written write-through by the combiners, read back by everyone)
    const bf16_t* W;        // [N, ldw] o_proj rows
    const bf16_t* res;      // [N] residual or null
    bf16_t* y;              // [N]
    int* cnt;               // [nkv + 1], ZERO on entry: cnt[hk] = finished slices of kv head hk, cnt[nkv] = combined kv heads
    int* err;               // set to 1 when a spin gives up
    const int* pos_dev;     // position of the token (device)
    int nh, group, nkv, smax, nsplit_cap;
    int N, K, ldw;          // K = nh * 128
    float scale_log2e;
    int wdelay;             // s_sleep(16) periods the weight request waits at t = 0, so that the attention's K / V requests reach the memory first
    long long* stamps;      // lab: [grid][16] s_memrealtime stamps (100 MHz) of the phases, or null
};
#define AO_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define AO_STAMP1(i) do { if (p.stamps && threadIdx.x == 256) p.stamps[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define ATTN_OPROJ_DYN_LDS (68 * 1024)     // dynamic LDS request (K * 2 bytes used): with the ~16 KiB static part, more than half a CU's LDS -> ONE workgroup per CU
#define ATTN_OPROJ_SPIN (1 << 22)

// PH = bit mask of the phases a launch runs: 1 = attention slices, 2 = combine, 4 = projection.  Hardware: PH = 7, ONE launch of
// ceil(N / 16) workgroups of 512 threads (one per CU; the grid must be co-resident: the host takes this path only when
// ceil(N / 16) <= the CU count).  The CPU emulator (tests/emu: one workgroup at a time) runs the phases as three launches, PH = 1, 2, 4.
//
//   t = 0   every wave that has no attention task requests its two o_proj rows (sixteen 16-B loads per lane) -- 33.5 MB at 7B widths,
//           the whole matrix, in flight in registers chip-wide.
//   phase 1 attention task s = (slice, head block, kv head) runs on the first four waves of workgroup s % G (tasks beyond G: next round):
//           K / V rows requested first (row `pos` patched from the roped k_new / v_new afterwards), q roped into LDS, the owner of `pos`
//           appends k_new / v_new to the cache; attn_slice_compute = attn_decode_kernel's arithmetic; partials stored write-through, drained,
//           one arrival on cnt[hk].  Then these waves request their o_proj rows too.
//   phase 2 the LAST nkv workgroups are the combiners: workgroup G - nkv + hk waits for cnt[hk] = slices x head blocks, combines the
//           `group` heads of kv head hk (attn_combine_head = attn_decode_combine_kernel's arithmetic, partials read past the L1), stores
//           o write-through, drains, one arrival on cnt[nkv].
//   phase 4 everyone waits for cnt[nkv] = nkv, reads o (8 KB, agent-scope loads) into LDS, dot products, residual, store.
// Hand-offs: guide G16 "sc1 stores and sc1 loads both sides" + a relaxed agent-scope counter polled by one lane; every spin is bounded.
template <int PH>
__global__ __launch_bounds__(512) void attn_oproj_kernel(AttnOprojArgs p) {
#pragma clang fp reassociate(off)
    constexpr int HD = 128, HALF = 64;
    constexpr bool ALL = PH == 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    __shared__ AttnSliceSmem sm2[1];
    __shared__ __attribute__((aligned(16))) bf16_t knew[HD], vnew[HD];
    __shared__ float wgtf[4][COMBINE_CHUNK];
    __shared__ float redf[4][2];
    bf16_t* xs = (bf16_t*)vl2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = tid >> 8, lt = tid & 255, lwave = lt >> 6, kl = lane & 15;
    const int G = gridDim.x, b = blockIdx.x;
    const int nvec = p.K >> 3;
    const int row0 = b * 16 + wave * 2;
    const int pos = p.pos_dev[0];
    // a position at or beyond the cache end (device-side position of a replayed graph that ran past the cache): the step's result is
    // discarded by the host; touch nothing and wait for nobody
    if (pos >= p.smax) return;
    const int ctx = pos + 1, group = p.group;
    const int nblk = (group + 3) >> 2, nslice = (ctx + 63) >> 6;
    const int per_head = nslice * nblk, ntask = per_head * p.nkv;
    u32x4 wv[2][8];
    float resv[2] = {0.f, 0.f};
    auto issue_w = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (row0 + r < p.N) {
                if (p.res) resv[r] = bf2f(p.res[row0 + r]);
                const bf16_t* wp = p.W + (size_t)(row0 + r) * p.ldw;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int v = i * 64 + lane;
                    if (v < nvec) wv[r][i] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)v * 8));
                }
            }
        }
    };
    // attention task s = (slice, head block, kv head) runs on workgroup s % G in round s / G, on the workgroup's FIRST four waves; its last
    // four waves only keep the barrier count (five per round) -- so that a wave's registers hold either a slice or the o_proj rows, never both
    const int myrounds = b < ntask ? (ntask - b + G - 1) / G : 0;
    AO_STAMP(0);
    // (a wave's loads return in order: a wave that polls a counter or combines must not have its weight rows in flight in front of that)
    const bool combiner = b >= G - p.nkv;
    if (ALL && !combiner && (myrounds == 0 || half == 1)) {
        for (int i = 0; i < p.wdelay; ++i) __builtin_amdgcn_s_sleep(16);
        issue_w();
    }
    // ---- phase 1: attention tasks
    if ((PH & 1) && myrounds > 0) {
        if (half == 1) {
            for (int it = 0; it < 5 * myrounds; ++it) __syncthreads();
        } else {
            AttnSliceSmem& sm = sm2[0];
            const float* cp = p.cos_t + (size_t)pos * HALF;
            const float* sp = p.sin_t + (size_t)pos * HALF;
            for (int it = 0; it < myrounds; ++it) {
                const int s = b + G * it;
                const int hk = s % p.nkv, split = (s / p.nkv) / nblk, h0 = ((s / p.nkv) % nblk) * 4;
                const int ng = group - h0 < 4 ? group - h0 : 4;
                const int k0 = split * 64;
                bf16_t* Kb = p.kcache + (size_t)hk * p.smax * HD;
                bf16_t* Vb = p.vcache + (size_t)hk * p.smax * HD;
                // K / V rows of the slice (keys beyond the context read row ctx - 1 = `pos`, which is not in the cache yet: patched below)
                uint32_t vv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    int kk = k0 + lwave * 16 + i;
                    kk = kk < ctx ? kk : ctx - 1;
                    vv[i] = *(const uint32_t*)(Vb + (size_t)kk * HD + lane * 2);
                }
                const int key = k0 + lwave * 16 + kl;
                const bool valid = key < ctx;
                const int krow = valid ? key : ctx - 1;
                const bf16_t* kr = Kb + (size_t)krow * HD;
                u32x4 kreg[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) kreg[c] = *(const u32x4*)(kr + c * 8);
                const bool owner = pos >= k0 && pos < k0 + 64;
                for (int u = lt; u < ng * HALF; u += 256) {
                    const int h = u / HALF, d = u % HALF;
                    const bf16_t* qh = p.qkv + (size_t)(hk * group + h0 + h) * HD;
                    float o1, o2;
                    rope_pair(bf2f(qh[d]), bf2f(qh[d + HALF]), cp[d], sp[d], o1, o2);
                    sm.qs[h][d] = bf2f(f2bf(o1));
                    sm.qs[h][d + HALF] = bf2f(f2bf(o2));
                }
                if (owner && lt < HALF) {
                    const bf16_t* kn = p.qkv + (size_t)(p.nh + hk) * HD;
                    const bf16_t* vn = p.qkv + (size_t)(p.nh + p.nkv + hk) * HD;
                    float o1, o2;
                    rope_pair(bf2f(kn[lt]), bf2f(kn[lt + HALF]), cp[lt], sp[lt], o1, o2);
                    const bf16_t k1 = f2bf(o1), k2 = f2bf(o2), v1 = vn[lt], v2 = vn[lt + HALF];
                    knew[lt] = k1; knew[lt + HALF] = k2; vnew[lt] = v1; vnew[lt + HALF] = v2;
                    Kb[(size_t)pos * HD + lt] = k1;            // DynamicCache.update: with two head blocks both owners write the same bytes
                    Kb[(size_t)pos * HD + lt + HALF] = k2;
                    Vb[(size_t)pos * HD + lt] = v1;
                    Vb[(size_t)pos * HD + lt + HALF] = v2;
                }
                __syncthreads();
                AO_STAMP(1);
                if (owner) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int kk = k0 + lwave * 16 + i;
                        if ((kk < ctx ? kk : ctx - 1) == pos) vv[i] = *(const uint32_t*)(&vnew[lane * 2]);
                    }
                    if (krow == pos) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) kreg[c] = *(const u32x4*)(&knew[c * 8]);
                    }
                }
                attn_slice_compute<true>(sm, lt, kreg, vv, valid, ng, p.scale_log2e,
                                         p.partial + ((size_t)(hk * group + h0) * p.nsplit_cap + split) * 130, (size_t)p.nsplit_cap * 130, true);
                AO_STAMP(2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores
                __syncthreads();
                AO_STAMP(3);
                if (lt == 0) __hip_atomic_fetch_add(p.cnt + hk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            AO_STAMP(4);
            if (ALL && !combiner) issue_w();
        }
    }
    // ---- phase 2: the combiners
    if ((PH & 2) && combiner) {
        const int hk = b - (G - p.nkv);
        if (ALL) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(p.cnt + hk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per_head) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > ATTN_OPROJ_SPIN) { *p.err = 1; break; }
                }
            }
            __syncthreads();
        }
        AO_STAMP(5);
        const int q4 = tid >> 7, d = tid & 127;
        const int live = nslice < p.nsplit_cap ? nslice : p.nsplit_cap;
        for (int hp = 0; hp < group; hp += 4) {
            const bool act = hp + q4 < group;
            const int head = hk * group + (act ? hp + q4 : group - 1);
            attn_combine_head<true, 128, true>(p.partial + (size_t)head * p.nsplit_cap * 130, live, d, wgtf[q4], redf[q4], p.o + (size_t)head * HD, act);
            __syncthreads();                                        // the weights / (M, 1/L) slots are reused by the next four
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.cnt + p.nkv, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        AO_STAMP(6);
        if (ALL) issue_w();
    }
    // ---- phase 4: the projection
    if (PH & 4) {
        if (ALL) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(p.cnt + p.nkv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.nkv) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > ATTN_OPROJ_SPIN) { *p.err = 1; break; }
                }
            }
            __syncthreads();
        } else {
            issue_w();
        }
        AO_STAMP(7);
        for (int i = tid; i < p.K / 4; i += 512)
            *(uint64_t*)(xs + 4 * i) = __hip_atomic_load((const uint64_t*)p.o + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        AO_STAMP(8);
        float a[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = i * 64 + lane;
            if (v < nvec) {
                const u32x4 xv = *(const u32x4*)(xs + (size_t)v * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[0] = dot2_bf16(wv[0][i][q], xv[q], a[0]);
                    a[1] = dot2_bf16(wv[1][i][q], xv[q], a[1]);
                }
            }
        }
        a[0] = wave_sum(a[0]);
        a[1] = wave_sum(a[1]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (row0 + r < p.N) {
                    float o = a[r];
                    if (p.res) o += resv[r];
                    p.y[row0 + r] = f2bf(o);
                }
            }
        }
        AO_STAMP(9);
        AO_STAMP1(10);
    }
}

// Sampled next token: temperature -> top-k -> top-p -> softmax -> one draw, on the fp32 logits of the decode step, ONE workgroup.
//
// Replaces what HF `GenerationMixin._sample` does with torch ops when the reference calls `model.generate(..., do_sample=True, temperature=,
// top_p=)` (videollama2/__init__.py:93-106; the gradio demo uses temperature 0.2): the logits warpers of HF:generation/logits_process.py
//   TemperatureLogitsWarper   scores / temperature
//   TopKLogitsWarper          remove scores < the k-th largest (ties with it stay)                      (generation_config default top_k = 50)
//   TopPLogitsWarper          sort ascending, softmax, cumsum; remove where cumsum <= 1 - top_p; the largest always stays
// then probs = softmax(scores) and torch.multinomial(probs, 1).  The draw itself cannot be the reference's bit for bit (torch.multinomial
// consumes its Philox stream in an implementation-defined way), so the contract is: the KEPT SET and the renormalised probabilities are the
// warpers', and the token is the inverse CDF of those probabilities, in token-index order, at the uniform number `u` the host hands over
// (oracle/sampling_oracle.py restates exactly this and is pinned to the live HF warpers).
// ONE documented deviation: top-p is a KEY threshold here -- every token whose score EQUALS the boundary score is kept -- while HF's
// TopPLogitsWarper cuts INSIDE a group of exactly tied scores (its sort orders the tie arbitrarily and removes the part whose cumulative probability is
// still <= 1 - top_p).  With exactly tied logits at the boundary (possible after temperature scaling of low-precision logits) the kept set is a superset
// of HF's by members of that one tie group, each of the boundary probability; HF's own choice among them is an artefact of its sort's tie order, not a
// defined semantics, so the oracle masks the boundary group (`boundary_tokens`) when it compares kept sets.
//
// No sort: floats are compared through their order-preserving 32-bit keys, and both thresholds are found by radix descent over the key bits,
// 8 bits per pass -- counts for top-k, probability MASS for top-p.  Mass is summed in 2^-40 fixed point (64-bit integer LDS atomics): integer
// addition is associative, so every histogram is the same whatever order the lanes arrive in, and the kernel is deterministic.  The logits
// (128 KB at V = 32000, 600 KB at 152064) stay in L2; a pass is V / 1024 coalesced loads per thread.
#pragma once
#include "dev_common.h"

__device__ __forceinline__ uint32_t sample_key(float x) {          // monotonic: a < b  <=>  key(a) < key(b)  (no NaNs in logits)
    const uint32_t b = __builtin_bit_cast(uint32_t, x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// `scores / temperature` as torch computes it (correctly rounded fp32 division) whatever -ffast-math turns a float division into: the quotient of
// two floats formed in double and rounded once more is the correctly rounded float quotient (53 >= 2 * 24 + 2 bits: the double rounding is innocuous)
#define VL2_FDIV_RN(a, b) ((float)((double)(a) / (double)(b)))
#define SAMPLE_FIX 1099511627776.0f                                // 2^40: a probability relative to the maximum, as a 64-bit integer

struct SampleArgs {
    const float* logits;   // [V] fp32
    int V;
    float temperature;     // > 0
    int top_k;             // 0 (or >= V) = off
    float top_p;           // >= 1 = off
    const float* u;        // uniform numbers in [0, 1): u[step] is consumed
    int* tok;              // -> the sampled token
    int* hist;             // optional: hist[step] = token
    int step;              // used when state == null
    int* state;            // optional (hipGraph-replayable decode): step = state[1]; afterwards state[0] (position) and state[1] advance by one
    float* dbg;            // optional [4]: {kept tokens, kept mass / total mass, threshold score, max score} of the call
};

// inclusive prefix sum of a 64-bit value over the lanes of a wave (six shuffle steps)
__device__ __forceinline__ unsigned long long sample_wave_scan(unsigned long long v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl(v, lane >= d ? lane - d : lane);
        if (lane >= d) v += t;
    }
    return v;
}
// The bin of a 256-bin histogram at which a running total crosses a threshold, found by ONE wave (lane L owns the bins of scan positions 4 L .. 4 L + 3;
// a serial walk by one thread was ~100 LDS round trips per level and most of the kernel's time).  Scan order: ascending bins (DESC = false) or from bin
// 255 downwards (DESC = true).  Crossing rule: DESC: the first position with acc + h >= thr (top-k: the rank is inside this bin); ascending: the first with
// acc + h > thr (top-p: the cumulative mass exceeds the threshold).  No crossing: the last position.  Results (the bin, and acc in FRONT of it) -> *bin, *acc.
template <bool DESC>
__device__ __forceinline__ void sample_find_bin(const unsigned long long* hist, unsigned long long base, unsigned long long thr, int lane,
                                                unsigned* bin, unsigned long long* acc_out, bool* found) {
    unsigned long long h[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int o = 4 * lane + j; h[j] = hist[DESC ? 255 - o : o]; mine += h[j]; }
    const unsigned long long incl = base + sample_wave_scan(mine, lane), excl = incl - mine;
    const bool cross = DESC ? (excl < thr && incl >= thr) : (excl <= thr && incl > thr);
    const bool any = !__all(!cross);
    *found = any ? cross : lane == 63;
    if (*found) {
        unsigned long long a = excl;
        int j = 0;
        for (; j < 3; ++j) {
            if (DESC ? (a + h[j] >= thr) : (a + h[j] > thr)) break;
            a += h[j];
        }
        const int o = 4 * lane + j;
        *bin = (unsigned)(DESC ? 255 - o : o);
        *acc_out = a;
    }
}

// CACHE: the scaled scores live in dynamic LDS (V * 4 bytes, V <= 32768: the 32000-token vocabularies) after pass A, so the ten later passes read LDS
// instead of re-reading the logits from L2 and re-dividing them (153 -> ~60 us per token at V = 32000); larger vocabularies stream from L2.
template <bool CACHE>
__global__ __launch_bounds__(1024) void sample_token_kernel(SampleArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    float* sc = (float*)vl2_smem;
    auto score = [&](int i) -> float { return CACHE ? sc[i] : VL2_FDIV_RN(p.logits[i], p.temperature); };
    __shared__ unsigned long long hist64[256];
    __shared__ float s_red[16];
    __shared__ uint32_t s_prefix;
    __shared__ unsigned long long s_base, s_target, s_thr;
    __shared__ int s_sel, s_tok, s_kept;
    __shared__ unsigned long long s_m[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = p.V;
    const float T = p.temperature;
    if (tid == 0) { s_kept = 0; s_tok = -1; }
    // ---- pass A: the maximum score (the softmax shift; also the last key of every descent)
    float mx = -3.4e38f;
    for (int i = tid; i < V; i += 1024) {
        const float s0 = VL2_FDIV_RN(p.logits[i], T);
        if (CACHE) sc[i] = s0;
        mx = fmaxf(mx, s0);
    }
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, s_red[w]);
    __syncthreads();
    // ---- top-k: the key of the k-th largest score by radix descent over COUNTS (from the top bin downwards)
    uint32_t key_lo = 0;                                            // kept: key >= key_lo
    if (p.top_k > 0 && p.top_k < V) {
        uint32_t prefix = 0;
        unsigned long long need = (unsigned long long)p.top_k;     // rank still to be found inside the current prefix
        for (int lvl = 3; lvl >= 0; --lvl) {
            if (tid < 256) hist64[tid] = 0ull;
            __syncthreads();
            const int sh = lvl * 8;
            // (a thread's elements mostly fall into the same bin at the upper levels -- the top byte of a key is sign + exponent -- so runs of equal
            //  bins are summed in registers and flushed once: the LDS atomics on two or three hot bins were most of the kernel's time)
            unsigned run_bin = 256u;
            unsigned long long run = 0ull;
            for (int i = tid; i < V; i += 1024) {
                const uint32_t k = sample_key(score(i));
                if (lvl == 3 || (k >> (sh + 8)) == (prefix >> (sh + 8))) {
                    const unsigned bin = (k >> sh) & 255u;
                    if (bin != run_bin) { if (run) atomicAdd(&hist64[run_bin], run); run_bin = bin; run = 0ull; }
                    run += 1ull;
                }
            }
            if (run) atomicAdd(&hist64[run_bin], run);
            __syncthreads();
            if (wave == 0) {
                unsigned d = 0;
                unsigned long long acc = 0;
                bool found = false;
                sample_find_bin<true>(hist64, 0ull, need, lane, &d, &acc, &found);
                if (found) { s_prefix = prefix | ((uint32_t)d << sh); s_base = need - acc; }      // exactly one lane
            }
            __syncthreads();
            prefix = s_prefix;
            need = s_base;
            __syncthreads();
        }
        key_lo = prefix;
    }
    // ---- top-p: ascending cumulative mass; removed while cumulative <= (1 - top_p) * Z.  Level 3 also yields Z (the total of its bins).
    unsigned long long Z = 0;
    {
        uint32_t prefix = 0;
        unsigned long long below = 0;                              // mass of the survivors with keys below the current prefix range
        const bool want_p = p.top_p < 1.0f;
        for (int lvl = 3; lvl >= (want_p ? 0 : 3); --lvl) {
            if (tid < 256) hist64[tid] = 0ull;
            __syncthreads();
            const int sh = lvl * 8;
            unsigned run_bin = 256u;
            unsigned long long run = 0ull;
            for (int i = tid; i < V; i += 1024) {
                const float s = score(i);
                const uint32_t k = sample_key(s);
                if (k >= key_lo && (lvl == 3 || (k >> (sh + 8)) == (prefix >> (sh + 8)))) {
                    const unsigned bin = (k >> sh) & 255u;
                    if (bin != run_bin) { if (run) atomicAdd(&hist64[run_bin], run); run_bin = bin; run = 0ull; }
                    run += (unsigned long long)(__expf(s - mx) * SAMPLE_FIX);
                }
            }
            if (run) atomicAdd(&hist64[run_bin], run);
            __syncthreads();
            if (wave == 0) {
                unsigned long long r = s_thr;
                if (lvl == 3) {
                    unsigned long long z = hist64[4 * lane] + hist64[4 * lane + 1] + hist64[4 * lane + 2] + hist64[4 * lane + 3];
                    z = __shfl(sample_wave_scan(z, lane), 63);
                    // (1 - top_p) in fp32 as HF forms it, times Z in double: the largest integer mass that still counts as "<= threshold"
                    r = want_p ? (unsigned long long)((double)(1.0f - p.top_p) * (double)z) : 0ull;
                    if (lane == 0) { s_target = z; s_thr = r; }
                }
                unsigned d = 0;
                unsigned long long acc = 0;
                bool found = false;
                sample_find_bin<false>(hist64, below, r, lane, &d, &acc, &found);     // the first bin whose cumulative mass exceeds the threshold: (partly) kept
                if (found) { s_prefix = prefix | ((uint32_t)d << sh); s_base = acc; }
            }
            __syncthreads();
            if (lvl == 3) Z = s_target;
            prefix = s_prefix;
            below = s_base;
            __syncthreads();
        }
        if (want_p && prefix > key_lo) key_lo = prefix;             // kept: cumulative mass up to and including the key exceeds the threshold
    }
    // ---- the draw: inverse CDF over the kept tokens in INDEX order.  Thread t owns the contiguous chunk [t * c, (t + 1) * c)
    const int c = (V + 1023) / 1024;
    const int i0 = tid * c, i1 = i0 + c < V ? i0 + c : V;
    unsigned long long mine = 0;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) {
        const float s = score(i);
        if (sample_key(s) >= key_lo) { mine += (unsigned long long)(__expf(s - mx) * SAMPLE_FIX); ++cnt; }
    }
    if (cnt) atomicAdd(&s_kept, cnt);
    // (every wave scans its 64 chunk masses with shuffles, thread 0 walks the 16 wave totals: a serial walk over 1024 LDS words was ~50 us)
    const unsigned long long incl_w = sample_wave_scan(mine, lane);
    if (lane == 63) s_m[wave] = incl_w;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < 16; ++w) tot += s_m[w];
        const int step = p.state ? p.state[1] : p.step;
        float u = p.u[step];
        u = u < 0.f ? 0.f : (u >= 1.f ? 0.99999994f : u);
        unsigned long long tgt = (unsigned long long)((double)u * (double)tot);
        if (tgt >= tot) tgt = tot - 1;
        unsigned long long acc = 0;
        int w = 0;
        for (; w < 15; ++w) {
            if (tgt < acc + s_m[w]) break;
            acc += s_m[w];
        }
        s_sel = w;                                                   // the owning wave; the exclusive prefix of its first chunk in s_base
        s_base = acc;
        s_target = tgt;
        if (p.dbg) { p.dbg[0] = (float)s_kept; p.dbg[1] = Z ? (float)((double)tot / (double)Z) : 1.f; p.dbg[3] = mx; }
    }
    __syncthreads();
    const unsigned long long excl_t = s_base + incl_w - mine;
    // the one chunk whose interval holds the target (a wave's last non-empty chunk if rounding left the target at its very end)
    const bool own = wave == s_sel && mine > 0 && excl_t <= s_target && (s_target < excl_t + mine);
    if (own) {
        const unsigned long long tgt = s_target;
        unsigned long long acc = excl_t;
        int sel = -1, last = -1;
        for (int i = i0; i < i1; ++i) {
            const float s = score(i);
            if (sample_key(s) >= key_lo) {
                const unsigned long long f = (unsigned long long)(__expf(s - mx) * SAMPLE_FIX);
                if (f > 0) last = i;
                if (f > 0 && tgt < acc + f) { sel = i; break; }
                acc += f;
            }
        }
        s_tok = sel >= 0 ? sel : last;
    }
    __syncthreads();
    if (tid == 0) {
        const int sel = s_tok;
        *p.tok = sel;
        if (p.dbg) {
            const uint32_t kb = (key_lo & 0x80000000u) ? (key_lo & 0x7fffffffu) : ~key_lo;
            p.dbg[2] = key_lo ? __builtin_bit_cast(float, kb) : -3.4e38f;
        }
        if (p.state) {
            if (p.hist) p.hist[p.state[1]] = sel;
            p.state[0] += 1;
            p.state[1] += 1;
        } else if (p.hist) {
            p.hist[p.step] = sel;
        }
    }
}

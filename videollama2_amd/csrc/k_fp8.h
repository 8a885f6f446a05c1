// fp8 weights for the HBM-bound decode path (the fp8 half of BASELINE.json configs[4]; the audio half has no source in the
// reference tree).  Format: OCP e4m3fn (what gfx950's conversion / MFMA instructions read -- NOT MI300X's fnuz), ONE fp32 scale per output
// row: W[n][k] ~= scale[n] * q[n][k].  Decode streams every weight once per token and is bound by that stream (61 % of
// 8 TB/s with bf16 weights), so halving the bytes is the lever; the activations stay in the 16-bit element type (W8A16), products and sums
// are fp32 exactly as in gemv_bf16_kernel.  The reference has no fp8 path ("needs a calibration story the reference does not have",
// SURVEY 8f-5): the quantiser below IS the definition, restated in oracle/fp8_oracle.py and held bit for bit against it.
//
//   quant_fp8_rows_kernel : row n of a 16-bit weight matrix -> scale[n] = 2^e, e the smallest integer with max|w[n][:]| <= 448 * 2^e (a
//                           power of two, as the hardware's own MX block scales are: applying and removing it is EXACT, so the quantiser is
//                           reproducible bit for bit on any machine -- no division, whose fast-math lowering on the GPU is a reciprocal);
//                           q = RNE_e4m3fn(w * 2^-e).  A zero (or denormal-only) row gets scale 1.
//   gemv_fp8_kernel       : y = scale * (q x) (+ bias) (+ res) for one token; fused RMSNorm prologue, SwiGLU pairing and residual epilogue
//                           as gemv_bf16_kernel (k_decode.h), x staged in LDS in the 16-bit element type exactly as there
#pragma once
#include "k_decode.h"

typedef float vl2_f32x2_t __attribute__((ext_vector_type(2)));
#ifndef VL2_CVT_PK_F32_FP8                      // two e4m3fn bytes of a dword -> two floats (v_cvt_pk_f32_fp8; the CPU test build defines its own)
#define VL2_CVT_PK_F32_FP8(w, hi) __builtin_amdgcn_cvt_pk_f32_fp8((int)(w), (hi))
#define VL2_CVT_PK_FP8_F32(a, b, old, hi) __builtin_amdgcn_cvt_pk_fp8_f32((a), (b), (old), (hi))
#endif

// grid = N rows, block 256.  w [N, ldw] 16-bit elements, q [N, K] bytes (K % 16 == 0), scale [N] fp32.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ w, uint8_t* __restrict__ q, float* __restrict__ scale,
                                                             int K, long ldw) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* row = w + (size_t)blockIdx.x * ldw;
    float amax = 0.f;
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
        unpack8(*(const u32x4*)(row + k), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
    amax = wave_max(amax);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // 448 = 1.75 * 2^8: amax = 1.f * 2^x  ->  e = x - 8 (+1 if 1.f > 1.75), clamped; scale = 2^e and 2^-e are exact floats
    const unsigned ab = __builtin_bit_cast(unsigned, amax);
    const int E = (int)((ab >> 23) & 0xffu);
    int e = E == 0 ? 0 : (E - 127) - 8 + ((ab & 0x7fffffu) > 0x600000u ? 1 : 0);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float sc = __builtin_bit_cast(float, (unsigned)(127 + e) << 23), inv = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
    if (tid == 0) scale[blockIdx.x] = sc;
    uint8_t* qrow = q + (size_t)blockIdx.x * K;
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
        unpack8(*(const u32x4*)(row + k), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * inv;
        int lo = 0, hi = 0;
        lo = VL2_CVT_PK_FP8_F32(v[0], v[1], lo, false);
        lo = VL2_CVT_PK_FP8_F32(v[2], v[3], lo, true);
        hi = VL2_CVT_PK_FP8_F32(v[4], v[5], hi, false);
        hi = VL2_CVT_PK_FP8_F32(v[6], v[7], hi, true);
        *(u32x2*)(qrow + k) = u32x2{(unsigned)lo, (unsigned)hi};
    }
}

// ---- fp8 ACTIVATIONS for the prefill GEMMs on the fp8 matrix pipe (VL2_GEMM_FP8: v_mfma_f32_32x32x64_f8f6f4, k_gemm.h gemm3 / gemm4 FP8):
// row m of a 16-bit activation -> e4m3fn bytes with the same power-of-two row scale rule as the weights (sa[m] = 2^e, e the smallest integer
// with max|x[m][:]| <= 448 * 2^e; exact to apply and to remove), and the row's entry of the GEMM's row table (GemmArgs.row_norm, [M][2]):
// (0, sa[m]) or, for an RMS-norm-carrying GEMM (HF:modeling_mistral.py:46-48 with the weight folded into W), (0, sa[m] * rsqrt(mean x^2 + eps))
// -- the quantiser reads the whole row anyway, so the norm needs no statistics from the producer.  grid = M rows, block 256, K % 16 == 0.
__global__ __launch_bounds__(256) void quant_act_fp8_kernel(const bf16_t* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq,
                                                            float* __restrict__ rowtab, int K, int rms, float eps) {
#pragma clang fp reassociate(off)
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* row = x + (size_t)blockIdx.x * ldx;
    float amax = 0.f, ss = 0.f;
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
        unpack8(*(const u32x4*)(row + k), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { amax = fmaxf(amax, fabsf(v[j])); ss = __builtin_fmaf(v[j], v[j], ss); }
    }
    amax = wave_max(amax);
    ss = wave_sum(ss);
    if (lane == 0) { red[wave] = amax; red[4 + wave] = ss; }
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    ss = (red[4] + red[5]) + (red[6] + red[7]);
    const unsigned ab = __builtin_bit_cast(unsigned, amax);             // the scale exponent: quant_fp8_rows_kernel's bit arithmetic
    const int E = (int)((ab >> 23) & 0xffu);
    int e = E == 0 ? 0 : (E - 127) - 8 + ((ab & 0x7fffffu) > 0x600000u ? 1 : 0);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float sc = __builtin_bit_cast(float, (unsigned)(127 + e) << 23), inv = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
    if (tid == 0) {
        rowtab[2 * (size_t)blockIdx.x] = 0.f;
        rowtab[2 * (size_t)blockIdx.x + 1] = rms ? sc * rsqrtf(ss / (float)K + eps) : sc;
    }
    uint8_t* qrow = q + (size_t)blockIdx.x * ldq;
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
        unpack8(*(const u32x4*)(row + k), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * inv;
        int lo = 0, hi = 0;
        lo = VL2_CVT_PK_FP8_F32(v[0], v[1], lo, false);
        lo = VL2_CVT_PK_FP8_F32(v[2], v[3], lo, true);
        hi = VL2_CVT_PK_FP8_F32(v[4], v[5], hi, false);
        hi = VL2_CVT_PK_FP8_F32(v[6], v[7], hi, true);
        *(u32x2*)(qrow + k) = u32x2{(unsigned)lo, (unsigned)hi};
    }
}

struct Gemv8Args {
    const uint8_t* W;       // [N, ldw] e4m3fn bytes (SWIGLU: packed blocks of 64 rows = 32 gate rows then 32 up rows, like the 16-bit layout)
    const float* scale;     // [N] one per (packed) row
    const bf16_t* x;        // [K]
    const float* norm_w;    // fused RMSNorm prologue on x (or null)
    const bf16_t* res;      // [N_out] residual (or null)
    void* y;                // 16-bit or fp32 [N_out]
    int N, K, ldw;
    float eps;
    const float* bias;      // [N_out] or null; not with SWIGLU
    int rms_plain;          // RMS-normalise x without a weight vector (folded into W before quantisation)
};

// A wave owns NP PAIRS of weight rows per trip and keeps all their loads in flight: 8 x 16 B per lane and row array = 16 KiB per wave,
// the 16-bit kernel's SwiGLU footprint -- the stream is bound by bytes in flight, so rows of K <= 4096 (four 16-B vectors per lane) run
// two pairs at a time (NP = 2), longer rows one.  A pair = (gate j, up j) -> one output with SWIGLU, rows (2 jp, 2 jp + 1) -> two outputs
// without (N even).  x lives in LDS in the 16-bit element type exactly as gemv_bf16_kernel stages it; a weight dword becomes two packed
// element pairs (v_cvt_pk_f32_fp8 + the pack: every e4m3fn value is exact in bf16 and in half) and meets x in v_dot2: fp32 sums.
// grid = ceil(n_pairs / (4 NP)), block 256; dynamic LDS = K * 2 bytes.  K % 16 == 0, K <= 32704; NP = 2 needs K <= 4096.
template <bool SWIGLU, bool OUT_F32, int NP>
__global__ __launch_bounds__(256) void gemv_fp8_kernel(Gemv8Args p) {
#pragma clang fp reassociate(off)                  // the RMSNorm arithmetic in gemv_bf16_kernel's order: the staged x is the same bits
    extern __shared__ __attribute__((aligned(16))) unsigned char vl2_smem[];
    __shared__ float red[8];
    bf16_t* xs = (bf16_t*)vl2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pairs = p.N / 2;
    const int nvec = p.K >> 4;                       // 16-B vectors (16 weights) per row
    constexpr int SL = 8 / NP;                       // load slots per pair and pass
    const bool one_pass = nvec <= 64 * SL;
    u32x4 wv[8], uv[8];
    const int jp0 = (blockIdx.x * 4 + wave) * NP;    // this wave's first pair
    auto rows_of = [&](int jp, int& r0, int& r1) {
        r0 = SWIGLU ? (jp >> 5) * 64 + (jp & 31) : 2 * jp;
        r1 = SWIGLU ? r0 + 32 : r0 + 1;
    };
    auto issue_rows = [&](int v0) {
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            int r0, r1;
            rows_of(jp0 + pr, r0, r1);
            const uint8_t* w0p = p.W + (size_t)r0 * p.ldw;
            const uint8_t* w1p = p.W + (size_t)r1 * p.ldw;
            const bool live = jp0 + pr < n_pairs;
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int v = v0 + i * 64 + lane;
                if (live && v < nvec) {
                    wv[pr * SL + i] = __builtin_nontemporal_load((const u32x4*)(w0p + (size_t)v * 16));
                    uv[pr * SL + i] = __builtin_nontemporal_load((const u32x4*)(w1p + (size_t)v * 16));
                }
            }
        }
    };
    if (one_pass && jp0 < n_pairs) issue_rows(0);    // the weights do not depend on x: their latency overlaps the staging of x
    float rstd = 1.f;
    const bool norm = p.norm_w != nullptr || p.rms_plain;
    if (norm) {
        float ss = 0.f;
        for (int k = tid * 8; k < p.K; k += 2048) {
            float v[8];
            unpack8(*(const u32x4*)(p.x + k), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(v[j], v[j], ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)p.K + p.eps);
    }
    for (int k = tid * 8; k < p.K; k += 2048) {
        u32x4 raw = *(const u32x4*)(p.x + k);
        if (norm) {
            float v[8];
            unpack8(raw, v);
            f32x4 w0 = {1.f, 1.f, 1.f, 1.f}, w1 = w0;
            if (p.norm_w) { w0 = *(const f32x4*)(p.norm_w + k); w1 = *(const f32x4*)(p.norm_w + k + 4); }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (v[j] * rstd) * (j < 4 ? w0[j] : w1[j - 4]);
            raw = pack8(v);                          // HF: the norm's output is a 16-bit tensor
        }
        *(u32x4*)(xs + k) = raw;
    }
    __syncthreads();
    if (jp0 >= n_pairs) return;
    float a0[NP], a1[NP];
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) a0[pr] = a1[pr] = 0.f;
    for (int v0 = 0; v0 < nvec; v0 += 64 * SL) {
        if (!one_pass) issue_rows(v0);
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            if (jp0 + pr >= n_pairs) continue;
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int v = v0 + i * 64 + lane;
                if (v < nvec) {
                    const u32x4 x0 = *(const u32x4*)(xs + (size_t)v * 16), x1 = *(const u32x4*)(xs + (size_t)v * 16 + 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned xl = q < 2 ? x0[2 * q] : x1[2 * q - 4], xh = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
                        const vl2_f32x2_t w0l = VL2_CVT_PK_F32_FP8(wv[pr * SL + i][q], false), w0h = VL2_CVT_PK_F32_FP8(wv[pr * SL + i][q], true);
                        const vl2_f32x2_t w1l = VL2_CVT_PK_F32_FP8(uv[pr * SL + i][q], false), w1h = VL2_CVT_PK_F32_FP8(uv[pr * SL + i][q], true);
                        a0[pr] = dot2_bf16(pack2bf(w0l[0], w0l[1]), xl, a0[pr]);
                        a1[pr] = dot2_bf16(pack2bf(w1l[0], w1l[1]), xl, a1[pr]);
                        a0[pr] = dot2_bf16(pack2bf(w0h[0], w0h[1]), xh, a0[pr]);
                        a1[pr] = dot2_bf16(pack2bf(w1h[0], w1h[1]), xh, a1[pr]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
        const int jp = jp0 + pr;
        if (jp >= n_pairs) continue;
        int r0, r1;
        rows_of(jp, r0, r1);
        float s0 = wave_sum(a0[pr]), s1 = wave_sum(a1[pr]);
        if (lane == 0) {
            s0 *= p.scale[r0];
            s1 *= p.scale[r1];
            if (SWIGLU) {
                float o = silu_f(s0) * s1;
                if (p.res) o += bf2f(p.res[jp]);
                ((bf16_t*)p.y)[jp] = f2bf(o);
            } else {
                if (p.bias) { s0 += p.bias[r0]; s1 += p.bias[r1]; }
                if (p.res) { s0 += bf2f(p.res[r0]); s1 += bf2f(p.res[r1]); }
                if (OUT_F32) { ((float*)p.y)[r0] = s0; ((float*)p.y)[r1] = s1; }
                else { ((bf16_t*)p.y)[r0] = f2bf(s0); ((bf16_t*)p.y)[r1] = f2bf(s1); }
            }
        }
    }
}

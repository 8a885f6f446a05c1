// Device-side helpers shared by every gfx950 kernel in this library (wave = 64 lanes).
// No HIP runtime headers are included here: the translation unit (vl2_abi.hip) includes <hip/hip_runtime.h> first.
#pragma once
#include <stdint.h>

typedef uint16_t bf16_t;                                           // raw bits of one 16-bit element (bfloat16; IEEE half with -DVL2_ELEM_F16)
typedef short bf16x8 __attribute__((ext_vector_type(8)));          // 8 bf16 = one MFMA A/B fragment = 16 B
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define VL2_WAVE 64

// ---- the 16-bit element type of activations and weights.  Default build: bfloat16 (BASELINE.json configs[1]).  -DVL2_ELEM_F16 builds the
// same library on IEEE half -- the reference's own dtype (videollama2/__init__.py:60 `.half().cuda()`, model/__init__.py:71
// torch_dtype=float16): mfma_f32_32x32x16_f16 runs at the bf16 rate on gfx950 and carries three more mantissa bits.  Everything below this
// block is written against these helpers (`bf16_t` = "raw 16-bit element" in both builds; names kept from the bf16-only rounds).
#ifdef VL2_ELEM_F16
typedef _Float16 vl2_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 vl2_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf2f(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ float bf2f_s(short b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }           // round-to-nearest-even
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vl2_f16x2));
}
__device__ __forceinline__ float e_lo(uint32_t u) { return (float)__builtin_bit_cast(vl2_f16x2, u)[0]; }    // element 0 / 1 of a packed pair
__device__ __forceinline__ float e_hi(uint32_t u) { return (float)__builtin_bit_cast(vl2_f16x2, u)[1]; }
#define VL2_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vl2_f16x8, a), __builtin_bit_cast(vl2_f16x8, b), c, 0, 0, 0)
#define VL2_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vl2_f16x8, a), __builtin_bit_cast(vl2_f16x8, b), c, 0, 0, 0)
#define VL2_ELEM_NAME "fp16"
#else
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, ((uint32_t)b) << 16); }
__device__ __forceinline__ float bf2f_s(short b) { return __builtin_bit_cast(float, ((uint32_t)(uint16_t)b) << 16); }
// float -> bf16, round-to-nearest-even (torch semantics); on gfx950 this is v_cvt_pk_bf16_f32
typedef __bf16 bf16v2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16v2_hw));
}
__device__ __forceinline__ float e_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }              // element 0 / 1 of a packed pair
__device__ __forceinline__ float e_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
#define VL2_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define VL2_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define VL2_ELEM_NAME "bf16"
#endif

// ---- fp8 (OCP e4m3fn) operands on the matrix pipe: v_mfma_f32_32x32x64_f8f6f4, 64 k per instruction = twice the bf16 rate per byte of
// operand (MI355X_MICROARCH.md: 4.6 PF measured on the MX form).  The builtin is the block-scaled one with zero scale operands, which the
// backend selects as the unscaled instruction (as composable_kernel's intrin_mfma_f32_32x32x64f8f6f4 does).  A lane supplies 32 bytes per
// operand = two 16-B fragments as the 16-bit kernels read them (k chunks c and c + 2 of a 64-byte row slab for lane half c): the k index
// function of the instruction is the same for A and B, and the dot product does not care which k a byte is, so the 16-bit kernels' LDS
// image and fragment reads carry fp8 rows unchanged -- a row of 2 K' bytes is a row of K' 16-bit "elements".
typedef int vl2_i32x8 __attribute__((ext_vector_type(8)));
typedef int vl2_i32x4 __attribute__((ext_vector_type(4)));
#ifndef VL2_MFMA32_F8
#define VL2_MFMA32_F8(a_lo, a_hi, b_lo, b_hi, c)                                                                                                  \
    __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                                                                              \
        __builtin_shufflevector(__builtin_bit_cast(vl2_i32x4, a_lo), __builtin_bit_cast(vl2_i32x4, a_hi), 0, 1, 2, 3, 4, 5, 6, 7),                \
        __builtin_shufflevector(__builtin_bit_cast(vl2_i32x4, b_lo), __builtin_bit_cast(vl2_i32x4, b_hi), 0, 1, 2, 3, 4, 5, 6, 7), c, 0, 0, 0, 0, 0, 0)
#endif

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = e_lo(v[i]);
        f[2 * i + 1] = e_hi(v[i]);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// Sum over the 8 lanes of an aligned lane octet, result in every lane of the octet: three DPP adds (v_add_f32_dpp on the VALU;
// `__shfl_xor` would go through the LDS crossbar as ds_bpermute).  Order: (i, 7-i) pairs (row_half_mirror), then the quad's
// neighbours [1,0,3,2], then its halves [2,3,0,1] -- fixed, so every user gets the same bits.
__device__ __forceinline__ float octet_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// LDS-DMA, 16 B per lane: LDS destination = wave-uniform `lds_wave_base` + lane*16, global source per lane.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// nn.GELU(approximate="tanh") = HF `gelu_pytorch_tanh` (SigLIP MLP): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))),
// tanh(u) = 1 - 2 / (1 + e^{2u})
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (2.0f - 2.0f / (1.0f + __expf(2.0f * u)));
}

// XCD-aware, bijective remap of a 1-D block id: block b runs on XCD b%8 (observed); give every XCD one
// contiguous chunk of the logical tile order so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

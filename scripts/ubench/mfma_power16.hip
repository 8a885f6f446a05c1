// Micro-benchmark (diagnostic, not part of the library): the MFMA-only ceiling on random operands for the TWO bf16 MFMA shapes of gfx950 --
// v_mfma_f32_32x32x16_bf16 (this library's; 16 accumulator registers per instruction) and v_mfma_f32_16x16x32_bf16 (the vendor kernels' "MI16x16x1";
// 4 accumulator registers, half the accumulator traffic per MAC, twice the operand reads) -- same FLOPs per wave, operands rotated as in mfma_power.hip
// mode 1.  Round 5 question: is part of the vendor GEMMs' 20 % margin the instruction's energy?  Waves per CU: 8 (2 per SIMD) and 4 (1 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const bf16x8* __restrict__ src, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(i * 2 + 0) * 512 + threadIdx.x]; b[i] = src[(i * 2 + 1) * 512 + threadIdx.x]; }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int t = 0; t < iters; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[j], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        f32x4 acc[16];                                        // the same 64 accumulator registers, 16 blocks of 16 x 16
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int t = 0; t < iters; ++t) {
#pragma unroll
            for (int q = 0; q < 2; ++q)                       // 32 instructions of 16384 FLOP = the 16 x 32768 FLOP of the other shape
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(q * 2 + (j >> 3)) & 3], b[j & 3], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int SHAPE, int THREADS>
void run(const char* name, const bf16x8* src, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256, reps = 20;
    hipLaunchKernelGGL((k<SHAPE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, src, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<SHAPE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, src, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flop = (double)blocks * THREADS / 64 * iters * 16 * 32768.0;
    const double tf = flop / (ms * 1e-3) / 1e12;
    printf("%-64s %8.1f us  %7.1f TFLOP/s\n", name, ms * 1e3, tf);
}
int main() {
    const int n = 8 * 512 * 8;
    short* h = (short*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
        float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        unsigned u; __builtin_memcpy(&u, &g, 4); h[i] = (short)(u >> 16);
    }
    bf16x8* src; float* d; hipMalloc(&src, n * 2); hipMalloc(&d, 4096);
    hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 512>("32x32x16, random rotating operands, 8 waves per CU", src, d, 4000);
        run<16, 512>("16x16x32, random rotating operands, 8 waves per CU", src, d, 4000);
        run<32, 256>("32x32x16, random rotating operands, 4 waves per CU (1 per SIMD)", src, d, 8000);
        run<16, 256>("16x16x32, random rotating operands, 4 waves per CU (1 per SIMD)", src, d, 8000);
    }
    return 0;
}

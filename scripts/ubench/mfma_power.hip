// Micro-benchmark (diagnostic, not part of the library): the MFMA-only ceiling of the chip as a function of operand
// DATA.  Eight waves per CU issue v_mfma_f32_32x32x16_bf16 back to back from registers (no LDS, no memory) with
//   mode 0: constant operands (every lane the same value)      -> the clock-unconstrained number,
//   mode 1: N(0,1)-like random bf16 operands, 4 A x 4 B register sets rotated so consecutive MFMAs see new bits,
//   mode 2: random operands but the same A/B pair every MFMA (operand latches do not toggle).
// Under the board power limit the clock drops with switching activity, so mode 1 is the ceiling a real GEMM can
// approach; a kernel's "fraction of 2.5 PF" should be read against it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ src, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        if (MODE == 0) { for (int j = 0; j < 8; ++j) { a[i][j] = (short)0x3f80; b[i][j] = (short)0x3f00; } }
        else { a[i] = src[(i * 2 + 0) * 512 + threadIdx.x]; b[i] = src[(i * 2 + 1) * 512 + threadIdx.x]; }
    }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE == 2 ? a[0] : a[s], MODE == 2 ? b[0] : b[j], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, const bf16x8* src, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256, threads = 512, reps = 20;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, src, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, src, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double mfma = (double)blocks * threads / 64 * iters * 16;
    const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-44s %8.1f us  %7.1f TFLOP/s  => %.2f GHz if the pipe never idles\n", name, ms * 1e3, tf, tf / 2500.0 * 2.4);
}
int main() {
    const int n = 8 * 512 * 8;
    short* h = (short*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) {               // Box-Muller N(0,1) -> bf16 bits (truncate)
        float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
        float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        unsigned u; __builtin_memcpy(&u, &g, 4); h[i] = (short)(u >> 16);
    }
    bf16x8* src; float* d; hipMalloc(&src, n * 2); hipMalloc(&d, 4096);
    hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    const int iters = 4000;                      // ~0.5 ms per launch, 20 launches: long enough for the clock to settle
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("constant operands", src, d, iters);
        run<2>("random operands, same A/B every MFMA", src, d, iters);
        run<1>("random operands, rotating 4 A x 4 B sets", src, d, iters);
    }
    return 0;
}

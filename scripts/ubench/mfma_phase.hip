// Micro-benchmark (diagnostic, not part of the library): how fast do MFMA phases run under different wave-group
// schedules?  mode 0: all 8 waves issue MFMAs continuously (no barriers); mode 1: ping-pong, group = wave>>2;
// mode 2: ping-pong, group = wave&1; mode 3: all 8 waves, barrier every 16 MFMAs; mode 4: 4-wave blocks, no barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = MODE == 1 ? (wave >> 2) : (wave & 1);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)(0x3f00 + i); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto burst = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    };
    if (MODE == 0 || MODE == 4) {
        for (int t = 0; t < iters; ++t) burst();
    } else if (MODE == 3) {
        for (int t = 0; t < iters; ++t) { burst(); __builtin_amdgcn_s_barrier(); }
    } else {
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int t = 0; t < iters; ++t) {
            __builtin_amdgcn_s_barrier();          // "LOAD" phase: nothing to do
            __builtin_amdgcn_s_setprio(1);
            burst();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads, int iters) {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double waves = (double)blocks * threads / 64, mfma = waves * iters * 16;
    printf("%-34s %8.1f us  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD-slot @2.0GHz)\n", name, ms * 1e3, mfma * 32768.0 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.0e9 / (iters * 16.0) / ((MODE == 1 || MODE == 2) ? 2.0 : (threads == 512 ? 2.0 : 1.0)) * 2.0 / 2.0);
    hipFree(d);
}
int main() {
    const int iters = 2000;
    run<0>("8 waves, continuous", 512, iters);
    run<4>("4 waves, continuous", 256, iters);
    run<3>("8 waves, barrier / 16 MFMA", 512, iters);
    run<1>("ping-pong grp=wave>>2", 512, iters);
    run<2>("ping-pong grp=wave&1", 512, iters);
    return 0;
}

// Issue rate of v_exp_f32 against v_fma_f32 on gfx950: one wave per SIMD (256 threads, one workgroup), 8 independent chains,
// cycles per instruction from s_memtime.  build: scripts/ubench/build_lab.sh valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cyc, int iters) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = -0.001f * (threadIdx.x + i + 1);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]));
            if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// MFMA 32x32x16 bf16 (four independent accumulators) with NV independent v_fma_f32 (or v_exp_f32 when TR) between consecutive MFMAs:
// do the vector instructions issue in the MFMA's shadow (time stays at the MFMA rate) or do the two add up?
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8v __attribute__((ext_vector_type(8)));
template <int NV, bool TR>
__global__ __launch_bounds__(256) void mix_kernel(float* out, long long* cyc, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8v a = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80}, b = a;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = -0.001f * (threadIdx.x + i + 1);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (TR) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i * NV + k) & 7]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(i * NV + k) & 7]));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV, bool TR> static void run_mix(float* out, long long* cyc, const char* what) {
    const int iters = 5000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((mix_kernel<NV, TR>), dim3(1), dim3(256), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); }
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("MFMA 32x32x16 + %d x %-10s between MFMAs: %.1f ticks per MFMA group\n", NV, what, (double)c / (iters * 4.0));
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char* names[4] = {"v_exp_f32", "v_fma_f32", "v_exp_f32 + v_fma_f32 (pair)", "v_rcp_f32"};
    for (int op = 0; op < 4; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            if (op == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
            if (op == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
            if (op == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
            if (op == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 8 * (op == 2 ? 1 : 1);
        printf("%-32s %.2f counter ticks per %s (one wave per SIMD, 8 independent chains)\n", names[op], (double)c / n, op == 2 ? "pair" : "instruction");
    }
    run_mix<0, false>(out, cyc, "-");
    run_mix<2, false>(out, cyc, "v_fma_f32");
    run_mix<4, false>(out, cyc, "v_fma_f32");
    run_mix<6, false>(out, cyc, "v_fma_f32");
    run_mix<8, false>(out, cyc, "v_fma_f32");
    run_mix<12, false>(out, cyc, "v_fma_f32");
    run_mix<4, true>(out, cyc, "v_exp_f32");
    run_mix<8, true>(out, cyc, "v_exp_f32");
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz (s_memtime / readcyclecounter ticks)\n", clk);
    return 0;
}

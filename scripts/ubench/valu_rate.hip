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
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz (s_memtime / readcyclecounter ticks)\n", clk);
    return 0;
}

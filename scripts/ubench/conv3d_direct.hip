// Measurement only (VERDICT r01 #12 / BASELINE.json north_star wording "the STC 3D-conv stays an LDS-tiled direct kernel"):
// a DIRECT LDS-tiled Conv3d(4096 -> 4096, kernel 2, stride 2, padding 1) on the vector ALU (v_dot2_f32_bf16), at the T = 16
// shape of the STC sampler (videollama2/model/projector.py:164-174: input [16, 24, 24, 4096] channels-last, output
// [9, 13, 13, 4096]), timed against nothing else here -- the product runs this convolution as an implicit GEMM on the matrix
// cores (csrc/k_gemm.h gathered-A form: 0.50 ms).  Not part of libvl2hip.so.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/conv3d_direct.hip -o scripts/ubench/conv3d_direct && scripts/ubench/conv3d_direct
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint16_t bf16_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));

static inline float bf2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

#ifndef CONV_T          // (smaller values: the host-emulator check of the indexing, see the end of this comment block)
#define CONV_T 16
#define CONV_HW 24
#define CONV_C 4096
#endif
constexpr int T = CONV_T, H = CONV_HW, W = CONV_HW, C = CONV_C, TO = T / 2 + 1, HO = H / 2 + 1, WO = W / 2 + 1, NOUT = TO * HO * WO;
constexpr int TP = 64, TC = 64, KC = 64, LDR = KC + 8;      // tile: 64 positions x 64 output channels, 64 input channels per step

// (by-value uint32 parameters: __builtin_bit_cast applied directly to an ext-vector ELEMENT lvalue is miscompiled by this clang -- zeros)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2_t, a), __builtin_bit_cast(bf16v2_t, b), c, false);
}

// x [T][H][W][C], w [C_out][8 taps][C_in] (tap = kt*4 + kh*2 + kw), y [NOUT][C_out] fp32
__global__ __launch_bounds__(256) void conv3d_direct_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) bf16_t As[TP][LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Ws[TC][LDR];
    const int tid = threadIdx.x, p0 = blockIdx.x * TP, c0 = blockIdx.y * TC;
    const int tp = tid >> 4, tc = tid & 15;                 // thread: positions tp*4..+3, channels tc*4..+3
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // staging role: row = tid / 4 (64 rows), 16-element (32 B) quarter = tid % 4 -> two 16-B loads per thread per tile
    const int srow = tid >> 2, sq = (tid & 3) * 16;
    const int pos = p0 + srow;
    const int to = pos / (HO * WO), ho = (pos / WO) % HO, wo = pos % WO;
    for (int tap = 0; tap < 8; ++tap) {
        const int t = 2 * to - 1 + (tap >> 2), h = 2 * ho - 1 + ((tap >> 1) & 1), ww = 2 * wo - 1 + (tap & 1);
        const bool ok = pos < NOUT && t >= 0 && t < T && h >= 0 && h < H && ww >= 0 && ww < W;
        const bf16_t* xrow = x + ((size_t)(t * H + h) * W + ww) * C;
        const bf16_t* wrow = w + ((size_t)(c0 + srow) * 8 + tap) * C;
        for (int k0 = 0; k0 < C; k0 += KC) {
            u32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            if (ok) { a0 = *(const u32x4*)(xrow + k0 + sq); a1 = *(const u32x4*)(xrow + k0 + sq + 8); }
            const u32x4 w0 = *(const u32x4*)(wrow + k0 + sq), w1 = *(const u32x4*)(wrow + k0 + sq + 8);
            __syncthreads();
            *(u32x4*)&As[srow][sq] = a0; *(u32x4*)&As[srow][sq + 8] = a1;
            *(u32x4*)&Ws[srow][sq] = w0; *(u32x4*)&Ws[srow][sq + 8] = w1;
            __syncthreads();
#pragma unroll
            for (int k8 = 0; k8 < KC; k8 += 8) {
                u32x4 av[4], wv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = *(const u32x4*)&As[tp * 4 + i][k8];
#pragma unroll
                for (int j = 0; j < 4; ++j) wv[j] = *(const u32x4*)&Ws[tc * 4 + j][k8];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[i][j] = dot2_bf16(av[i][q], wv[j][q], acc[i][j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + tp * 4 + i;
        if (p < NOUT)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[(size_t)p * C + c0 + tc * 4 + j] = acc[i][j];
    }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main() {
    const size_t nx = (size_t)T * H * W * C, nw = (size_t)C * 8 * C, ny = (size_t)NOUT * C;
    std::vector<bf16_t> hx(nx), hw(nw);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hx) v = f2bf(rnd());
    for (auto& v : hw) v = f2bf(rnd() * 0.02f);
    bf16_t *dx, *dw; float* dy;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    dim3 grid((NOUT + TP - 1) / TP, C / TC);
    hipLaunchKernelGGL(conv3d_direct_kernel, grid, dim3(256), 0, 0, dx, dw, dy);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 5;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(conv3d_direct_kernel, grid, dim3(256), 0, 0, dx, dw, dy);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    // spot-check 48 outputs against a host loop (fp64 accumulation)
    std::vector<float> hy(ny);
    CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int n = 0; n < 48; ++n) {
        const int p = (n * 397 + 11) % NOUT, co = (n * 911 + 5) % C;
        const int to = p / (HO * WO), ho = (p / WO) % HO, wo = p % WO;
        double ref = 0;
        for (int tap = 0; tap < 8; ++tap) {
            const int t = 2 * to - 1 + (tap >> 2), h = 2 * ho - 1 + ((tap >> 1) & 1), ww = 2 * wo - 1 + (tap & 1);
            if (t < 0 || t >= T || h < 0 || h >= H || ww < 0 || ww >= W) continue;
            const bf16_t* xr = &hx[((size_t)(t * H + h) * W + ww) * C];
            const bf16_t* wr = &hw[((size_t)co * 8 + tap) * C];
            for (int k = 0; k < C; ++k) ref += (double)bf2f(xr[k]) * bf2f(wr[k]);
        }
        worst = fmax(worst, fabs(ref - hy[(size_t)p * C + co]) / (fabs(ref) + 1e-3));
    }
    const double flop = 2.0 * NOUT * (double)C * C * 8;
    printf("{\"kernel\": \"direct LDS-tiled Conv3d k2 s2 p1, v_dot2_f32_bf16 (vector ALU)\", \"shape\": \"[16,24,24,4096] -> [9,13,13,4096]\", "
           "\"ms\": %.3f, \"tflops\": %.1f, \"spot_check_max_rel_err\": %.2e, \"note\": \"product path = implicit GEMM on MFMA (gathered-A vl2_gemm)\"}\n",
           ms, flop / ms / 1e9, worst);
    return worst < 1e-3 ? 0 : 2;
}

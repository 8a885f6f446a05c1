// CLIP-ViT front end (HF:models/clip/modeling_clip.py CLIPVisionEmbeddings.forward):
//   patchify_kernel : frames [T,3,H,W] (fp32 / fp16 / bf16, NCHW as process_video returns them, mm_utils.py:199-201)
//                     -> im2col rows [T*G*G, Kp] bf16, k = c*P*P + ky*P + kx (the flatten order of the conv weight
//                     [D,3,P,P]); columns >= 3*P*P are zero (Kp = 640 for P = 14 keeps K a multiple of the GEMM's BK).
//                     The k=14,s=14 conv is then a GEMM whose epilogue adds position_embedding[1+p] and writes row
//                     t*(G*G+1)+1+p (GemmArgs out_grp / res_row_mod), i.e. torch.cat([cls, patches]) + pos without a pass.
//   fill_cls_kernel : row t*(G*G+1) = class_embedding + position_embedding[0]  (precomputed on the host once).
#pragma once
#include "dev_common.h"

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<_Float16>(const _Float16* p) { return (float)*p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t* p) { return __builtin_bit_cast(float, ((uint32_t)*p) << 16); }   // bfloat16 FRAMES (whatever the element type of the build)

// grid = (G, T); block 256
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ frames, bf16_t* __restrict__ out, int H, int W,
                                                       int P, int G, int Kp) {
    const int py = blockIdx.x, t = blockIdx.y;
    const int kvec = Kp >> 3, PP = P * P, Kreal = 3 * PP;
    const T* f = frames + (size_t)t * 3 * H * W;
    for (int e = threadIdx.x; e < G * kvec; e += 256) {
        const int px = e / kvec, k0 = (e - px * kvec) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            if (k < Kreal) {
                const int c = k / PP, r = k - c * PP, ky = r / P, kx = r - ky * P;
                v[j] = load_as_f32<T>(f + ((size_t)c * H + (py * P + ky)) * W + px * P + kx);
            } else {
                v[j] = 0.f;
            }
        }
        *(u32x4*)(out + ((size_t)(t * G + py) * G + px) * Kp + k0) = pack8(v);
    }
}

// uint8 ingest: frames [T,H,W,3] uint8 exactly as the decoder / resize step leaves them -> the same im2col
// rows, with the image processor's arithmetic tail (HF image_processing: x * rescale_factor, then (x - mean) / std, fp32)
// done in registers: half the bytes of bf16 frames over PCIe and HBM, no normalised copy on the host.
// nrm = {rescale, mean[3], 1/std[3]} (the division is a multiplication by the fp32 reciprocal: <= 1 fp32 ulp from the
// processor's, invisible after the bf16 rounding except on exact rounding ties).  grid = (G, T); block 256
struct U8Norm { float rescale, mean[3], inv_std[3]; };
__global__ __launch_bounds__(256) void patchify_u8_kernel(const unsigned char* __restrict__ frames, bf16_t* __restrict__ out,
                                                          int H, int W, int P, int G, int Kp, U8Norm nrm) {
    const int py = blockIdx.x, t = blockIdx.y;
    const int kvec = Kp >> 3, PP = P * P, Kreal = 3 * PP;
    const unsigned char* f = frames + (size_t)t * H * W * 3;
    for (int e = threadIdx.x; e < G * kvec; e += 256) {
        const int px = e / kvec, k0 = (e - px * kvec) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            if (k < Kreal) {
                const int c = k / PP, r = k - c * PP, ky = r / P, kx = r - ky * P;
                const float x = (float)f[((size_t)(py * P + ky) * W + px * P + kx) * 3 + c] * nrm.rescale;
                v[j] = (x - nrm.mean[c]) * nrm.inv_std[c];
            } else {
                v[j] = 0.f;
            }
        }
        *(u32x4*)(out + ((size_t)(t * G + py) * G + px) * Kp + k0) = pack8(v);
    }
}

// rows t*rows_per_frame of x [T*rows_per_frame, D] <- cls_pos [D];  grid = T, block 128 (D/8 <= 128... loop anyway)
__global__ __launch_bounds__(128) void fill_cls_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ cls_pos, int D,
                                                       int rows_per_frame) {
    bf16_t* dst = x + (size_t)blockIdx.x * rows_per_frame * D;
    for (int c = threadIdx.x * 8; c < D; c += 128 * 8) *(u32x4*)(dst + c) = *(const u32x4*)(cls_pos + c);
}

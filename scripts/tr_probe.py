#!/usr/bin/env python
"""Hardware check of the ds_read_b64_tr_b16 lane map the emulator (tests/emu/hip_emu.h) and csrc/k_attn2.h assume: compiles a
12-line kernel with hipcc at run time (GPU box only), fills LDS with lds[i] = i, lets lane l read from byte address 8 * l, and
prints whether lane l / element e received 64 (l >> 4) + 16 e + (l & 15) -- and the full table if not."""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void probe(short* out, const int* addr_bytes) {
    __shared__ __attribute__((aligned(16))) short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + addr_bytes[threadIdx.x]));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
extern "C" void run(short* out, const int* addr, void* stream) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, (hipStream_t)stream, out, addr); }
'''


def main():
    d = tempfile.mkdtemp()
    open(os.path.join(d, "p.hip"), "w").write(SRC)
    so = os.path.join(d, "p.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(d, "p.hip"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    ok_all = True
    for name, addr in (("linear 8*l", [8 * l for l in range(64)]),
                       ("attn2 V image (rows of 32 B, lane m -> row m>>2, col 4(m&3))", [(l >> 4) * 128 + ((l & 15) >> 2) * 32 + (l & 3) * 8 for l in range(64)])):
        a = torch.tensor(addr, dtype=torch.int32, device="cuda")
        lib.run(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        got = out.cpu().view(64, 4).tolist()
        # emulator model: lane (g, i), element e <- element (i & 3) of the 4 shorts loaded by lane 16 g + 4 e + (i >> 2)
        exp = [[addr[16 * (l >> 4) + 4 * e + ((l & 15) >> 2)] // 2 + (l & 3) for e in range(4)] for l in range(64)]
        ok = got == exp
        ok_all &= ok
        print(f"tr_probe [{name}]: {'matches the emulator model' if ok else 'DIFFERS'}")
        if not ok:
            for l in range(64):
                print(l, got[l], exp[l])
    sys.exit(0 if ok_all else 3)


if __name__ == "__main__":
    main()

#!/bin/bash
# round 6, call F: the vendor yardstick at the head (ours / hipBLASLt / ours on the 16x16x32 set, plain GEMMs, interleaved)
mkdir -p gpurun_out
timeout 600 python scripts/vendor_gemm_ref.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_vendor_gemm_ref.txt

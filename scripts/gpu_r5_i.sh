#!/bin/bash
# round 5, call I: gemm8 (k_gemm8.h: the 256 x 256 tile on four waves, 128 x 128 wave tiles) -- bit-identity on the device, then beside the library's
# own choice and the vendor GEMM on every shape of the step.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "pingpong_variant" -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_gemm8.log
timeout 600 python scripts/vendor_gemm_ref.py 3 2>&1 | grep -v amdgpu.ids | tee $O/vendor_gemm_ref.txt

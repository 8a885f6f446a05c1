#!/bin/bash
# round 6, call C: 16x16x32 mixed launch, 32-deep vs 64-deep big tiles; hardware tests of the flipped default
mkdir -p gpurun_out
timeout 300 python scripts/mix16_bench.py 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_mix16_bench_64deep.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_v21.py tests/test_gpu_api.py tests/test_gpu_tp.py -m gpu -x -q 2>&1 | tail -5

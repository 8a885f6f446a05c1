#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "patchify or layernorm" > gpurun_out/pytest_u8.log 2>&1; echo "pytest exit $?"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_bf16frames.json 2>/dev/null; echo $?
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --u8-frames > gpurun_out/bench_u8frames.json 2>/dev/null; echo $?

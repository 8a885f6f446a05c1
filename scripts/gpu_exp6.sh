#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
timeout 900 python scripts/kernel_bench.py --small --quick > gpurun_out/kernel_bench_small.txt 2>&1; echo "kb exit $?"
timeout 900 python scripts/shard_model.py > gpurun_out/shard_model.jsonl 2> gpurun_out/shard_model.err; echo "shard exit $?"

#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attn" > gpurun_out/pytest_attn.log 2>&1; echo "pytest exit $?"
timeout 600 python scripts/kernel_bench.py --quick 2>&1 | grep -i "attn" > gpurun_out/kb_attn.txt; echo "kb exit $?"
for T in 16 32; do timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames $T > gpurun_out/bench_a_T$T.json 2>/dev/null; done

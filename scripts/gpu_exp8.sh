#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; echo "bench exit $?"

#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python scripts/kernel_bench.py > gpurun_out/kernel_bench.log 2>&1; cat gpurun_out/kernel_bench.log | grep -v JSON
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.log; tail -3 gpurun_out/bench.err

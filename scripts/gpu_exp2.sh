#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "streams or golden_tower" > gpurun_out/pytest_streams.log 2>&1; echo "pytest exit $?"
for T in 8 16 32; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames $T > gpurun_out/bench_s3_T$T.json 2> gpurun_out/bench_s3_T$T.err; echo "T$T exit $?"
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames 32 --vit-streams 1 > gpurun_out/bench_s1_T32.json 2>/dev/null
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames 8 --vit-streams 1 > gpurun_out/bench_s1_T8.json 2>/dev/null

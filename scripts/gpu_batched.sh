#!/bin/bash
# batched-decode regression + throughput sweep
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "batched or skinny" > gpurun_out/pytest_batched.log 2>&1; echo "pytest exit $?"
for B in 4 8 16 32 64; do timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-batch $B 2>gpurun_out/bb_$B.err | python -c "import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $B, j.get('batched_decode'))"; done > gpurun_out/bench_batched.txt 2>&1

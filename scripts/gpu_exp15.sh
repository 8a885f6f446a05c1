#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "batched" > gpurun_out/pytest_batched.log 2>&1; echo "pytest exit $?"
for R in 1 2 4; do for B in 2 4; do timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-batch $B --tune 4=$R 2>/dev/null | python -c "import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rpw', $R, 'B', $B, j.get('batched_decode'))"; done; done > gpurun_out/bench_batched.txt 2>&1

#!/bin/bash
mkdir -p gpurun_out
for B in 2 4; do timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --prefill-batch $B 2>gpurun_out/bp_$B.err | python -c "import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $B, j['encode_ms'], j['prefill_ms'], j['forward_mfma_frac'], j.get('batched_prefill'))"; done > gpurun_out/bench_bprefill.txt 2>&1

#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "gemv or decode or golden or hipgraph or qwen2" > gpurun_out/pytest_dec.log 2>&1; echo "pytest exit $?"
timeout 600 python scripts/kernel_bench.py --quick 2>&1 | grep -i "gemv\|attn_decode" | head -3 > gpurun_out/kb_gemv.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dec.json 2>/dev/null; echo $?

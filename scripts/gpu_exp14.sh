#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -k "stc_direct or layernorm or 72b" > gpurun_out/pytest_72.log 2>&1; echo "pytest exit $?"
timeout 1500 python bench.py --model 72b --steps 3 --warmup 1 --new-tokens 16 --no-cpu-baseline > gpurun_out/bench_72b.json 2> gpurun_out/bench_72b.err; echo "72b exit $?"

#!/bin/bash
set -u
mkdir -p gpurun_out
./scripts/ubench/mfma_power > gpurun_out/mfma_power.txt 2>&1
for ns in 1 2 3 4; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --vit-streams $ns > gpurun_out/bench_vs$ns.json 2> gpurun_out/bench_vs$ns.err; echo "vs$ns exit $?"
done

#!/bin/bash
# Round 2, GPU call B: micro-benchmarks that steer the kernel work (norm-carrying GEMM costs, GEMM tile variants per shape).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 400 python scripts/gemm_norm_bench.py > $O/gemm_norm_bench.txt 2> $O/gemm_norm_bench.err
timeout 400 python scripts/kernel_bench.py --quick > $O/kernel_bench.txt 2> $O/kernel_bench.err
echo done

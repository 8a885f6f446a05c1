#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/kernel_bench.py --quick > gpurun_out/kernel_bench.log 2>&1; grep -v JSON gpurun_out/kernel_bench.log | head -30

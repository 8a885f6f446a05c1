#!/bin/bash
# round 5, call M: power / clock under each GEMM form (family 32x32x16, gemm9 16x16x32, four-wave gemm8, vendor)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05m; mkdir -p $O
timeout 400 python scripts/gemm_power_ab.py $O/gemm_power_ab.json --seconds 1.5 --reps 2 --forms v8,v16,v26,vendor > $O/gemm_power_ab.txt 2>&1; echo "rc $?"
cut -c1-300 $O/gemm_power_ab.txt

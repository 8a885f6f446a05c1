#!/bin/bash
# round 5, call N: s_memtime stamps inside gemm9's phases
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05n; mkdir -p $O
timeout 300 python scripts/gemm9_phase_stamps.py $O/gemm9_phase_stamps.json > $O/gemm9_phase_stamps.txt 2>&1; echo "rc $?"
cut -c1-400 $O/gemm9_phase_stamps.txt

#!/bin/bash
# Round 3: run a prebuilt C++ kernel lab binary (scripts/ubench/<name>, built here by scripts/ubench/build_lab.sh) on the GPU box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${2:-lab}
mkdir -p $O
timeout ${3:-300} scripts/ubench/${1:-gemm_lab} ${4:-3} > $O/${1:-gemm_lab}.txt 2> $O/${1:-gemm_lab}.err
echo rc=$? >> $O/${1:-gemm_lab}.err
tail -40 $O/${1:-gemm_lab}.txt
tail -20 $O/${1:-gemm_lab}.err

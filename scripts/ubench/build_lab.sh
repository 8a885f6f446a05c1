#!/bin/bash
# builds scripts/ubench/gemm_lab (and keeps the gfx950 .s next to it under /tmp/lab for inspection)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p /tmp/lab
cd /tmp/lab && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -I /root/repo/videollama2_amd/csrc /root/repo/scripts/ubench/${1:-gemm_lab}.hip -o /root/repo/scripts/ubench/${1:-gemm_lab} -save-temps 2>&1 | grep -v "warning: argument unused" | head -30

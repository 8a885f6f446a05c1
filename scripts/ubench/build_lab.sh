#!/bin/bash
# builds scripts/ubench/<name> from scripts/ubench/<src>.hip (gfx950 .s kept under /tmp/lab for inspection)
#   build_lab.sh [src [out [extra hipcc flags...]]]
SRC=${1:-gemm_lab}; OUT=${2:-$SRC}; shift; shift
mkdir -p /tmp/lab/$OUT && cd /tmp/lab/$OUT && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only "$@" -I /root/repo/videollama2_amd/csrc -I /root/repo/scripts/ubench /root/repo/scripts/ubench/$SRC.hip -o /root/repo/scripts/ubench/$OUT -save-temps 2>&1 | grep -v "warning: argument unused\|implicit conversion\|float u1\|~$\|warning generated" | head -30

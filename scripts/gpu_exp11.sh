#!/bin/bash
mkdir -p gpurun_out
for T in 8 32; do timeout 900 python scripts/kernel_bench.py --quick --frames $T > gpurun_out/kb_T$T.txt 2>&1; done

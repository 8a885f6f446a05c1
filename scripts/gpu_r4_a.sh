#!/bin/bash
# round 4, call B (lean hot loop): persistent GEMM (k_gemm6.h) in the C++ lab -- gemm4 (v8 LDS epilogue = bit reference, v9 C^T epilogue, v17 192-row tiles)
# against gemm6 (v60 persistent 256-row, v61 persistent 192-row, v62 192-row + two accumulator sets), then the default bench line
mkdir -p gpurun_out
export LAB_SHAPES=vit_qkv,vit_fc1,stc_s1,llm_qkv,llm_gateup,sq_4096,sq_8192x4096
timeout 300 scripts/ubench/gemm_lab 3 8,9,17,60,61,62 > gpurun_out/r04b_gemm_lab.txt 2> gpurun_out/r04b_gemm_lab.err
echo "lab rc=$?" >> gpurun_out/r04b_gemm_lab.txt
tail -30 gpurun_out/r04b_gemm_lab.txt
grep -v hash gpurun_out/r04b_gemm_lab.err | tail -20

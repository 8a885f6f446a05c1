#!/bin/bash
# round 4, call C: the persistent GEMM through the C ABI (auto = choose_gemm6) against variant 24 (= round 3's automatic choice), per shape
# of the T = 16 step + the ViT shapes on patch rows only; then the GEMM tests of the GPU suite and the default bench line
mkdir -p gpurun_out
timeout 600 python scripts/kernel_bench.py --frames 16 --peeled > gpurun_out/r04c_kernel_bench_T16.txt 2> gpurun_out/r04c_kernel_bench.err
tail -20 gpurun_out/r04c_kernel_bench_T16.txt; tail -5 gpurun_out/r04c_kernel_bench.err
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k gemm > gpurun_out/r04c_pytest_gemm.log 2>&1; tail -5 gpurun_out/r04c_pytest_gemm.log
timeout 600 python bench.py > gpurun_out/r04c_bench_T16.json 2> gpurun_out/r04c_bench.err; head -c 900 gpurun_out/r04c_bench_T16.json

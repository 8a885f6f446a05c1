#!/bin/bash
# Round 2, GPU call G: the GEMM round split (tests + per-shape A/B: auto vs forced kernels at T = 16 / 8 / 32), bench lines
# at T = 8 / 16 / 32 and for the VideoLLaMA2.1 family on the head of the round.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02g}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "gemm" > $O/pytest_gemm.log 2>&1
for T in 16 8 32; do timeout 400 python scripts/kernel_bench.py --frames $T > $O/kernel_bench_T$T.txt 2>&1; done
timeout 600 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
timeout 400 python bench.py --frames 8 --no-cpu-baseline > $O/bench_T8.json 2> $O/bench_T8.err
timeout 400 python bench.py --frames 32 --no-cpu-baseline > $O/bench_T32.json 2> $O/bench_T32.err
timeout 400 python bench.py --model v21 --no-cpu-baseline > $O/bench_v21_T16.json 2> $O/bench_v21.err
echo done

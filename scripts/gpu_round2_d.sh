#!/bin/bash
# Round 2, GPU call D: attention structures A/B (variants 1, 3, 4, 5), their GPU tests, the norm-chain micro-benchmark.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02d}
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "attn or norm_carrying" 2>&1 | tail -15 ) > $O/pytest_attn.log 2>&1
timeout 300 python scripts/attn_bench2.py > $O/attn_bench2.jsonl 2> $O/attn_bench2.err
timeout 400 python scripts/gemm_norm_bench.py > $O/gemm_norm_bench.txt 2> $O/gemm_norm_bench.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_T16.json 2> $O/bench_T16.err
echo done

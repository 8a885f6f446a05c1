#!/bin/bash
# Round 3, GPU call D: GEMM-touching GPU tests + per-shape kernel bench (auto vs forced) + default bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r03s}
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py tests/test_gpu_stages.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/pytest_ops.log 2>&1
grep -E "passed|failed|error" $O/pytest_ops.log | tail -3
timeout 600 python scripts/kernel_bench.py --frames 16 > $O/kernel_bench_T16.txt 2>&1
cat $O/kernel_bench_T16.txt | cut -c1-260
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline > $O/bench_T16_$i.json 2> $O/bench_T16_$i.err; python -c "
import json; j=json.loads(open('$O/bench_T16_$i.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('value','encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac')}, j['roofline']['achieved'], j['roofline']['dominant']['avg_launch_us'], j['vit_only']['ms'])"; done

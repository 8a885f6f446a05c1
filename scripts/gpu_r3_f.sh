#!/bin/bash
# Round 3, GPU call F: decode-touching GPU tests (graph == eager, stage == per-op, batched, continuous batching) + bench
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r03u}
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_api.py tests/test_gpu_v21.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/pytest_decode.log 2>&1
grep -E "passed|failed|error" $O/pytest_decode.log | tail -3
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline > $O/bench_T16_$i.json 2> $O/bench_T16_$i.err; python -c "
import json; j=json.loads(open('$O/bench_T16_$i.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac','forward_mfma_frac')})"; done

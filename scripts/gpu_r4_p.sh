#!/bin/bash
# round 4, call P: decode tail engine after the load-serialisation fix -- decode rate with / without it, alternating; then its GPU tests
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
for f in 0 16 0 16; do
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 64 --stage-flags $f --no-vit-only 2>> $O/r04p_bench.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage-flags $f', {k: j[k] for k in ('encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac')})"
done | tee $O/r04p_decode_ab.txt
( timeout 1200 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_v21.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/r04p_pytest.log 2>&1; tail -3 $O/r04p_pytest.log

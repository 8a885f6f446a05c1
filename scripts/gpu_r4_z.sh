#!/bin/bash
# round 4, call Z: the fp8 kernels in the fp16 build (GPU test + the bench line with both options on)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04z; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fp8.py -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_fp8.log 2>&1
grep -E "^\[fp8\]|passed|failed|Error|assert" $O/pytest_fp8.log | cut -c1-260 | tail -12
timeout 600 python bench.py --dtype fp16 --decode-weights fp8 --no-cpu-baseline --no-vit-only 2>$O/bench.err | tail -1 > $O/bench_fp16_fp8.json
python -c "
import json; j=json.loads(open('$O/bench_fp16_fp8.json').read().strip().splitlines()[-1]); print(j['dtype'], {k: j[k] for k in ('encode_ms','prefill_ms','decode_ms_per_token')}, (j.get('decode_fp8') or {}).get('ms_per_token'))"
tail -2 $O/bench.err

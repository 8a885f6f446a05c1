#!/bin/bash
# round 6, call D: the 160-row tile -- hardware bit-identity tests, kernel A/B, default bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python scripts/tile160_bench.py 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_tile160_bench.txt
timeout 600 python bench.py > gpurun_out/r06_bench_T16_runD.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_T16_runD.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac','decode_hbm_frac')}, 'vit', d['vit_only']['ms'], 'dominant', d['roofline']['dominant']['avg_launch_us'], d['roofline']['frac'])
for s in d['roofline']['shapes']: print(s)
PY

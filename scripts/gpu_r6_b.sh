#!/bin/bash
# round 6, call B: the 16x16x32 mixed launch (gate/up) -- hardware test, kernel A/B, pipeline A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py -m gpu -x -q -k "mfma16 or stage" 2>&1 | tail -5
timeout 300 python scripts/mix16_bench.py 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_mix16_bench.txt
for i in 1 2; do
  for fl in 0 32768; do
    timeout 300 python bench.py --steps 10 --warmup 3 --stage-flags $fl 2>/dev/null > gpurun_out/r06_bench_T16_flags${fl}_$i.json
    python - <<PY
import json
d=json.loads(open('gpurun_out/r06_bench_T16_flags${fl}_$i.json').read().strip().splitlines()[-1])
print('flags $fl run $i', {k:d[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac')}, 'dominant us', d['roofline']['dominant']['avg_launch_us'])
PY
  done
done

#!/bin/bash
# round 6, call E: the 16x16x32 set on the decoder's o / down projections -- hardware tests, kernel A/B, pipeline A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_v21.py tests/test_gpu_tp.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python scripts/mfma16_set_bench.py 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_mfma16_set_bench.txt
for i in 1 2; do
  for fl in 65536 0; do
    timeout 300 python bench.py --steps 10 --warmup 3 --stage-flags $fl --no-cpu-baseline --no-vit-only 2>/dev/null > gpurun_out/r06_bench_T16_mf16set_flags${fl}_$i.json
    python - <<PY
import json
d=json.loads(open('gpurun_out/r06_bench_T16_mf16set_flags${fl}_$i.json').read().strip().splitlines()[-1])
print('flags $fl run $i', {k:d[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac')}, [ (s['N'],s['K'],s['avg_launch_us']) for s in d['roofline']['shapes'][:6]])
PY
  done
done

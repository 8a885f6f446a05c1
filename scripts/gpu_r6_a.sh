#!/bin/bash
# round 6, call A: GPU tests at the head after the lab split / x-first GEMV / pinned attention arithmetic / loud vocabulary rows, then the default bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06_pytest_gpu_runA.log 2>&1
echo "pytest rc=$?"
tail -40 gpurun_out/r06_pytest_gpu_runA.log
timeout 600 python bench.py > gpurun_out/r06_bench_T16_runA.json 2> gpurun_out/r06_bench_T16_runA.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_T16_runA.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac','decode_hbm_frac')})
PY

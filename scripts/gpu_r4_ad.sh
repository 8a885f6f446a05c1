#!/bin/bash
# round 4, call AD: the driver's own commands at HEAD -- the full GPU suite with per-test durations, then `bench.py --gpus 1 --steps 20 --warmup 5`
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04ad; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=25 2>&1 ) > $O/pytest_gpu_durations.log 2>&1
grep -E "passed|failed|^real|s call|s setup" $O/pytest_gpu_durations.log | cut -c1-200 | tail -32
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err
python -c "
import json; j=json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('value','steps','warmup','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac','decode_hbm_frac')}, j['roofline']['frac'])"

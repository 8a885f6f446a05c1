#!/bin/bash
# round 5, call K: the sampling kernel with the scaled scores in LDS (tests + time per call), and the default line with this round's PMC traffic file.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_ops.py -x -q -s -k "sampl or pingpong or producer_side" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^\[sampling\]|passed|failed|rc " $O/pytest.log | cut -c1-300
cp gpurun_out/r05_sampling_kernel_us.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
python -c "
import json; j=json.loads(open('$O/bench_T16.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac','forward_mfma_frac')}, {k: j['roofline'][k] for k in ('frac','traffic','traffic_source')})"

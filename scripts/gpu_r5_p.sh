#!/bin/bash
# round 5, call P: last check of the library as committed: smoke(), GEMM-family + stage tests, the default bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05p; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py -x -q -k "gemm or stage_calls" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log | cut -c1-300
timeout 600 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
python -c "
import json; j=json.loads(open('$O/bench_T16.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac','forward_mfma_frac')}, j['roofline']['frac'], j['cpu_baseline'])"

#!/bin/bash
# Round 3: the full GPU suite + smoke() + the default bench line (what the driver runs at round end)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r03full}
mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|^\[rccl|^\[tp-local|passed|failed|error|FAILED|ERROR|^real" $O/pytest_gpu_full.log | tail -80 > $O/pytest_gpu.log
cp gpurun_out/r03_parity.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
tail -3 $O/smoke.log; grep -E "passed|failed|^real" $O/pytest_gpu.log | tail -3; python -c "
import json; j=json.loads(open('$O/bench_T16.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac','forward_mfma_frac')}, j['roofline']['frac'], j['cpu_baseline']['value'])"

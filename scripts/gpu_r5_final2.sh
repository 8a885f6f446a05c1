#!/bin/bash
# Round 5, evidence at the head after gemm9 went in: smoke(), the full GPU suite, the default bench line and its rocprofv3 kernel-trace summary.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05final2
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s --durations=15 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|^\[rccl|^\[tp-local|^\[fp8\]|passed|failed|error|FAILED|ERROR|^real" $O/pytest_gpu_full.log | cut -c1-400 | tail -170 > $O/pytest_gpu.log
grep -E "passed|failed|^real" $O/pytest_gpu.log | tail -3
cp gpurun_out/r05_parity.json $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
python -c "
import json; j=json.loads(open('$O/bench_T16.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac','forward_mfma_frac')}, j['roofline']['frac'])"

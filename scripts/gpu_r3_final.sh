#!/bin/bash
# Round 3, final evidence on the head of the round: smoke(), the full GPU suite, the default bench line, a rocprofv3 kernel trace
# of the same command, the connector's direct kernels, the other workload lines (T = 8 / 32, VideoLLaMA2.1, 72B), the shard model.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03final}
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|^\[rccl|^\[tp-local|passed|failed|error|FAILED|ERROR|^real" $O/pytest_gpu_full.log | tail -120 > $O/pytest_gpu.log
cp gpurun_out/r03_parity.json gpurun_out/r02_rccl_world1.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
timeout 300 python scripts/stc_bench.py > $O/stc_bench.txt 2>&1
timeout 300 python bench.py --frames 8 --no-cpu-baseline > $O/bench_T8.json 2> $O/bench_T8.err
timeout 400 python bench.py --frames 32 --no-cpu-baseline > $O/bench_T32.json 2> $O/bench_T32.err
timeout 600 python bench.py --model v21 --no-cpu-baseline > $O/bench_v21.json 2> $O/bench_v21.err
timeout 1200 python bench.py --model 72b --no-cpu-baseline > $O/bench_72b.json 2> $O/bench_72b.err
timeout 600 python scripts/shard_model.py --reps 3 > $O/shard_model.jsonl 2> $O/shard_model.err
timeout 600 python scripts/shard_model.py --reps 3 --splitk > $O/shard_model_splitk.jsonl 2> $O/shard_model_splitk.err
tail -2 $O/smoke.log; grep -E "passed|failed|^real" $O/pytest_gpu.log | tail -3
for f in T16 T8 T32 v21 72b; do python -c "
import json; j=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', {k: j[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac','forward_mfma_frac')}, j['roofline']['frac'])" 2>&1 | tail -1; done
tail -3 $O/shard_model.jsonl | cut -c1-400; tail -3 $O/shard_model_splitk.jsonl | cut -c1-400

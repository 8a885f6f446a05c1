#!/bin/bash
# Round 2, GPU call I (final evidence on the head of the round): smoke(), the full GPU suite, the default bench line, a rocprofv3
# kernel trace of the same command, the connector's direct kernels, and the 72B line.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02i}
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|^\[rccl|passed|failed|error|FAILED|ERROR" $O/pytest_gpu_full.log | tail -120 > $O/pytest_gpu.log
cp gpurun_out/r02_parity.json gpurun_out/r02_rccl_world1.json $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
timeout 300 python scripts/stc_bench.py > $O/stc_bench.txt 2>&1
timeout 1200 python bench.py --model 72b --no-cpu-baseline > $O/bench_72b.json 2> $O/bench_72b.err
echo done

#!/bin/bash
# Round 2, GPU call A: full GPU test-suite (incl. the configs[1] parity tests -> r02_parity.json), the default bench line,
# a rocprofv3 kernel-trace of the same command, and the RCCL world-1 smoke.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02a}
mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
nproc > $O/host_cores.txt; lscpu | head -20 >> $O/host_cores.txt
timeout 120 python scripts/tr_probe.py > $O/tr_probe.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|passed|failed|error|FAILED|ERROR" $O/pytest_gpu_full.log | tail -80 > $O/pytest_gpu.log
cp gpurun_out/r02_parity.json $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
timeout 300 python scripts/attn_bench2.py > $O/attn_bench2.jsonl 2> $O/attn_bench2.err
timeout 300 python scripts/rccl_world1.py > $O/rccl_world1.json 2> $O/rccl_world1.err
R=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
echo done

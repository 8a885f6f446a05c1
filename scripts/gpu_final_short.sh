#!/bin/bash
# Short evidence refresh (when GPU minutes are scarce): default bench line with cpu_baseline, T=8 / T=32 / v21 lines, rocprofv3
# kernel trace of the default bench, smoke.  The full set is gpu_final.sh.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"
timeout 300 python bench.py --frames 8 --no-cpu-baseline > gpurun_out/bench_T8.log 2>> gpurun_out/bench.err; echo "bench T8 exit $?"
timeout 300 python bench.py --frames 32 --no-cpu-baseline > gpurun_out/bench_T32.log 2>> gpurun_out/bench.err; echo "bench T32 exit $?"
timeout 300 python bench.py --model v21 --no-cpu-baseline > gpurun_out/bench_v21.log 2>> gpurun_out/bench.err; echo "bench v21 exit $?"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_bench.log 2>&1; echo "trace exit $?"
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/trace/bench_kernel_trace.csv
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_bench.log 2>&1; echo "trace exit $?"
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/trace/bench_kernel_trace.csv

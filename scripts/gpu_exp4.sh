#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python scripts/kernel_bench.py --small --quick > gpurun_out/kernel_bench_small.txt 2>&1; echo "kb exit $?"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shard_trace -o st -- python $GRAFT_REPO_ROOT/scripts/shard_model.py --frames 16 --worlds 8 --reps 3 > $GRAFT_REPO_ROOT/gpurun_out/shard_trace.log 2>&1; echo "trace exit $?"

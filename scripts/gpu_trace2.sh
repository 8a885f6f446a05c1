#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_now -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --new-tokens 2 > $GRAFT_REPO_ROOT/gpurun_out/trace_now.log 2>&1; echo "trace exit $?"

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_b16 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --decode-batch 16 --no-graph --new-tokens 16 > /dev/null 2>&1

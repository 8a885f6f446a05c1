#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; echo "bench exit $?"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_now -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --new-tokens 4 > $GRAFT_REPO_ROOT/gpurun_out/trace_now.log 2>&1; echo "trace exit $?"

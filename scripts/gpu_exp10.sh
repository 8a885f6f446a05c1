#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
timeout 900 python bench.py --model v21 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v21.json 2> gpurun_out/bench_v21.err; echo "v21 exit $?"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_v21 -o t -- python $GRAFT_REPO_ROOT/bench.py --model v21 --steps 3 --warmup 1 --no-cpu-baseline --new-tokens 2 > $GRAFT_REPO_ROOT/gpurun_out/trace_v21.log 2>&1; echo "trace exit $?"

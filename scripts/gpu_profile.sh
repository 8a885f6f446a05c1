#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace of the default bench + HBM-traffic PMC passes (separate passes).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_bench.log 2>&1
echo "trace exit $?"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$set -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/pmc_${set}_bench.log 2>&1
  echo "pmc $set exit $?"
done
cd $GRAFT_REPO_ROOT
ls gpurun_out/trace | head; 
# keep the merged payload small
find gpurun_out -name "*kernel_trace.csv" -size +30M -exec sh -c 'head -200000 "$1" > "$1.head"; rm "$1"' _ {} \;
du -sh gpurun_out

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic_$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_traffic_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/traffic_$c.log 2>&1; echo "pmc $c exit $?"
done

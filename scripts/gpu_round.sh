#!/bin/bash
# One gpurun call: parity tests + bench + rocprof kernel trace.  Everything judged later is copied to profiles/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(rocminfo 2>/dev/null | grep -m1 gfx9 || true)" | tee gpurun_out/host.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit: $?"; tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
echo "rocprof exit: $?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
# keep only the small summaries (the raw trace can be large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete

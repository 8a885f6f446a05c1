#!/bin/bash
# round 4, call AH: rocprofv3 kernel trace of the bench with the fp8-weights decode pass on (per-kernel durations of gemv_fp8_kernel inside the graph)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04ah; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --decode-weights fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_fp8.csv \;
rm -rf $O/trace
grep -E "gemv|attn_decode|argmax" $O/kernel_stats_bench_fp8.csv | cut -d, -f1-5 | cut -c1-160

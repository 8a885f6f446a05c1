#!/bin/bash
# rocprofv3 kernel trace of the bench command at HEAD (whole steps only) + the default bench line with cpu_baseline
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03trace}; mkdir -p $O; R=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
timeout 900 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
head -8 $O/kernel_stats_bench.csv | cut -c1-150; tail -1 $O/bench_T16.json | cut -c1-600

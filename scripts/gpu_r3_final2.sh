#!/bin/bash
# Round 3, second evidence call: HBM-traffic PMC passes of every GEMM shape of the step at HEAD (separate --pmc runs, kernel-trace
# only) and the rocprofv3 kernel trace of the bench command without the extra tower-only passes.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03final2}
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/traffic_$c -o pmc -- python $R/scripts/gemm_traffic_pmc.py > $R/$O/traffic_$c.log 2>&1 ); echo "pmc $c exit $?"
done
F=$(find $O/traffic_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/traffic_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python scripts/gemm_traffic_post.py $F $W > $O/r03_gemm_traffic.json 2> $O/traffic_post.err; tail -2 $O/traffic_post.err
rm -rf $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
head -5 $O/r03_gemm_traffic.json; head -4 $O/kernel_stats_bench.csv | cut -c1-160; tail -1 $O/trace_bench.log | cut -c1-300

#!/bin/bash
# kernel trace of the per-rank pieces of the sharded-connector cut at T = 16, R = 8 (2 frames per rank), one GPU
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r06_shard_prof
mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o sm -- python $R/scripts/shard_model.py --frames 16 --worlds 8 --reps 3 > $R/$O/shard_model.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
tail -2 $O/shard_model.log | cut -c1-400
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:40]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY

#!/bin/bash
# Round 3, GPU call A: kernel TIMELINE of one forward (encode + prefill + 1 decode token) to price gaps between kernels,
# plus the default bench line of the round-2 head on this box (baseline for every later A/B of the round).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03a}
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench_T16.json 2> $O/bench_T16.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o tl -- python $R/bench.py --steps 2 --warmup 1 --new-tokens 1 --no-graph --no-cpu-baseline > $R/$O/trace_bench.log 2>&1 )
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/timeline_kernel_trace.csv \;
rm -rf $O/trace
python - <<PY > $O/timeline_summary.txt 2>&1
import csv
rows = list(csv.DictReader(open("$O/timeline_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "kernels")
PY
echo done

#!/bin/bash
# Round 2, GPU call E: the full GPU suite on the head of the round (stage-level ABI, encoder graphs, TP decode graph, continuous
# batching), the default bench line, a rocprofv3 kernel trace of the same command, and the FETCH_SIZE / WRITE_SIZE passes
# (separate --pmc runs, kernel-trace only) for roofline.traffic.  Everything lands in gpurun_out/<dir>.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02e}
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|^\[rccl|passed|failed|error|FAILED|ERROR" $O/pytest_gpu_full.log | tail -80 > $O/pytest_gpu.log
cp gpurun_out/r02_parity.json gpurun_out/r02_rccl_world1.json $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/traffic_$c -o pmc -- python $R/scripts/gemm_traffic_pmc.py > $R/$O/traffic_$c.log 2>&1 )
done
F=$(find $O/traffic_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/traffic_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python scripts/gemm_traffic_post.py "$F" "$W" > $O/gemm_traffic.json 2> $O/gemm_traffic.err
cp "$F" $O/traffic_fetch_counters.csv; cp "$W" $O/traffic_write_counters.csv
rm -rf $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
timeout 120 scripts/ubench/conv3d_direct > $O/conv3d_direct.json 2>&1
timeout 300 python scripts/stc_bench.py > $O/stc_bench.txt 2>&1
echo done

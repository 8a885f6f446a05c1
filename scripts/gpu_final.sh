#!/bin/bash
# Round evidence: GPU parity tests, default bench line (with cpu_baseline), T=8 / T=32 lines, the VideoLLaMA2.1 line, the
# per-rank shard model, rocprofv3 kernel trace of the default bench, smoke.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"
timeout 900 python bench.py --frames 8 --no-cpu-baseline > gpurun_out/bench_T8.log 2>> gpurun_out/bench.err; echo "bench T8 exit $?"
timeout 900 python bench.py --frames 32 --no-cpu-baseline > gpurun_out/bench_T32.log 2>> gpurun_out/bench.err; echo "bench T32 exit $?"
timeout 900 python bench.py --model v21 --no-cpu-baseline > gpurun_out/bench_v21.log 2>> gpurun_out/bench.err; echo "bench v21 exit $?"
timeout 1500 python bench.py --model 72b --steps 3 --warmup 1 --new-tokens 16 --no-cpu-baseline > gpurun_out/bench_72b.log 2>> gpurun_out/bench.err; echo "bench 72b exit $?"
for B in 4 16 64; do timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-batch $B 2>> gpurun_out/bench.err | tail -1 > gpurun_out/bench_batched_$B.log; done; echo "batched done"
timeout 900 python scripts/shard_model.py > gpurun_out/shard_model.jsonl 2> gpurun_out/shard_model.err; echo "shard exit $?"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_bench.log 2>&1; echo "trace exit $?"
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/trace/bench_kernel_trace.csv
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

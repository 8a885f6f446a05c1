#!/bin/bash
# Round 6, final evidence at the head of the round: smoke(), the full GPU suite, the default bench line (roofline + cpu_baseline), the
# rocprofv3 kernel trace of the bench command, the fp16 / fp8-decode / T = 8 / 32 / VideoLLaMA2.1 / 72B lines, connector direct kernels,
# the shard model, and the multi-rank control flow (ranks sharing the one GPU over gloo; timings meaningless).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r06final}
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -s --durations=15 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|^\[rccl|^\[tp-local|^\[fp8\]|passed|failed|error|FAILED|ERROR|^real" $O/pytest_gpu_full.log | cut -c1-400 | tail -170 > $O/pytest_gpu.log
grep -E "passed|failed|^real" $O/pytest_gpu.log | tail -3
cp gpurun_out/r06_parity.json gpurun_out/r04_fp8_parity.json gpurun_out/r05_fp8_prefill_parity.json gpurun_out/r05_sampling_kernel_us.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/traffic_$c -o pmc -- python $R/scripts/gemm_traffic_pmc.py > $R/$O/traffic_$c.log 2>&1 ); echo "pmc $c exit $?"
done
F=$(find $O/traffic_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/traffic_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python scripts/gemm_traffic_post.py $F $W > $O/r06_gemm_traffic.json 2> $O/traffic_post.err; tail -2 $O/traffic_post.err
rm -rf $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_T16_driver_cmd.json 2> $O/bench_T16_driver_cmd.err
timeout 600 python bench.py --prefill-weights fp8 --decode-weights fp8 --no-cpu-baseline --no-vit-only > $O/bench_fp8_both.json 2> $O/bench_fp8_both.err
timeout 300 python scripts/stc_bench.py 2>&1 | grep -v amdgpu.ids > $O/stc_bench.txt
timeout 200 scripts/ubench/decode_lab 1650 > $O/decode_lab.txt 2>&1
timeout 600 python bench.py --dtype fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
timeout 300 python bench.py --frames 8 --no-cpu-baseline > $O/bench_T8.json 2> $O/bench_T8.err
timeout 400 python bench.py --frames 32 --no-cpu-baseline > $O/bench_T32.json 2> $O/bench_T32.err
timeout 600 python bench.py --model v21 --no-cpu-baseline > $O/bench_v21.json 2> $O/bench_v21.err
timeout 1200 python bench.py --model 72b --no-cpu-baseline > $O/bench_72b.json 2> $O/bench_72b.err
timeout 600 python scripts/shard_model.py --reps 3 > $O/shard_model.jsonl 2> $O/shard_model.err
export VL2_DIST_BACKEND=gloo
for N in 2 4 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) \
      bench.py --gpus $N --steps 2 --warmup 1 --new-tokens 8 > $O/bench_gloo_$N.json 2> $O/bench_gloo_$N.err
  echo "gloo N=$N exit $?"
done
unset VL2_DIST_BACKEND
for f in T16 T16_driver_cmd fp16 fp8_both T8 T32 v21 72b; do python -c "
import json; j=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', {k: j[k] for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac','forward_mfma_frac')}, j['roofline']['frac'], (j.get('decode_fp8') or {}).get('ms_per_token'), (j.get('prefill_fp8') or {}).get('prefill_ms'))" 2>&1 | tail -1; done
tail -2 $O/shard_model.jsonl | cut -c1-300

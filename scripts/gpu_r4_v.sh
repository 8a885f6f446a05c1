#!/bin/bash
# round 4, call V: the LLM prefill with its row_norm_finalize launches back (call U measured the accidental in-GEMM reduction: 26.0 ms):
# default bench line x2 + the A/B flag (stage flag 4 = in-GEMM reduction), rocprofv3 kernel trace of the bench, HBM-traffic PMC passes of
# every GEMM shape at HEAD (separate --pmc runs, kernel-trace only), the other workload lines (T = 8 / 32, VideoLLaMA2.1), the shard model.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04v; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
date +%s > $O/t0
( timeout 900 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/pytest_stage.log 2>&1; tail -2 $O/pytest_stage.log
timeout 900 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
for i in 1 2; do for fl in 0 4; do
  timeout 600 python bench.py --stage-flags $fl --no-cpu-baseline --no-vit-only --steps 5 --warmup 2 2>$O/bench_ab.err | tail -1 > $O/bench_f${fl}_$i.json
done; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/traffic_$c -o pmc -- python $R/scripts/gemm_traffic_pmc.py > $R/$O/traffic_$c.log 2>&1 ); echo "pmc $c exit $?"
done
F=$(find $O/traffic_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/traffic_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python scripts/gemm_traffic_post.py $F $W > $O/r04_gemm_traffic.json 2> $O/traffic_post.err; tail -2 $O/traffic_post.err
rm -rf $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE
timeout 300 python bench.py --frames 8 --no-cpu-baseline > $O/bench_T8.json 2> $O/bench_T8.err
timeout 400 python bench.py --frames 32 --no-cpu-baseline > $O/bench_T32.json 2> $O/bench_T32.err
timeout 600 python bench.py --model v21 --no-cpu-baseline > $O/bench_v21.json 2> $O/bench_v21.err
timeout 600 python scripts/shard_model.py --reps 3 > $O/shard_model.jsonl 2> $O/shard_model.err
date +%s > $O/t1
python - <<'PY'
import glob, json
for f in ["gpurun_out/r04v/bench_T16.json", "gpurun_out/r04v/bench_T8.json", "gpurun_out/r04v/bench_T32.json", "gpurun_out/r04v/bench_v21.json"] + sorted(glob.glob("gpurun_out/r04v/bench_f*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], r["dtype"], r["value"], "ms", r["ms_per_step"], "enc", r["encode_ms"], "pre", r["prefill_ms"], "dec", r["decode_ms_per_token"],
              "fwd", r.get("forward_mfma_frac"), "hbm", r.get("decode_hbm_frac"), "roof", r.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -c 600 $O/r04_gemm_traffic.json; echo; tail -4 $O/shard_model.jsonl | cut -c1-500

#!/bin/bash
# round 4, call S: the fp16 build on hardware -- full-depth parity in both element types (one fp32 truth), then the default bench line in each
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity_full.py::test_configs1_full_depth_end_to_end tests/test_gpu_parity_full.py::test_configs1_full_depth_end_to_end_fp16_build -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/r04s_pytest_fp16.log 2>&1
grep -E "parity-full|passed|failed|Error|assert" $O/r04s_pytest_fp16.log | cut -c1-220 | grep -v "decode step [0-9]* logits" | tail -30
cp $O/r04_parity.json $O/r04s_parity.json 2>/dev/null
for dt in bf16 fp16 bf16 fp16; do
  timeout 600 python bench.py --dtype $dt --no-cpu-baseline --steps 5 --warmup 2 2>$O/r04s_bench_$dt.err | tail -1 > $O/r04s_bench_$dt.$RANDOM.json
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r04s_bench_*.json")):
    try:
        r = json.loads(open(f).read())
        print(f.split("/")[-1], r["dtype"], r["value"], "ms", r["ms_per_step"], "enc", r["encode_ms"], "pre", r["prefill_ms"], "dec", r["decode_ms_per_token"], "roof", r.get("roofline", {}).get("achieved"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/r04s_bench_fp16.err

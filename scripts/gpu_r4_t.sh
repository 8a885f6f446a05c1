#!/bin/bash
# round 4, call T: STC strip depthwise kernel + fused SE (tests, stc_bench, pipeline A/B against the five-launch chain = --stage-flags 32),
# and the fp16 build's full-depth parity again after the frame-dtype fix
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "dwconv or stc or small_linear" 2>&1 ) > $O/r04t_pytest_stc.log 2>&1
tail -5 $O/r04t_pytest_stc.log | cut -c1-300
( timeout 600 python scripts/stc_bench.py 2>&1 ) > $O/r04t_stc_bench.txt; cat $O/r04t_stc_bench.txt | grep -v amdgpu.ids
for i in 1 2 3; do for fl in 32 0; do
  timeout 600 python bench.py --stage-flags $fl --no-cpu-baseline --no-vit-only --steps 5 --warmup 2 2>$O/r04t_bench.err | tail -1 > $O/r04t_bench_f${fl}_$i.json
done; done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r04t_bench_f*.json")):
    try:
        r = json.loads(open(f).read())
        print(f.split("/")[-1], r["value"], "ms", r["ms_per_step"], "enc", r["encode_ms"], "pre", r["prefill_ms"], "dec", r["decode_ms_per_token"], {k: v for k, v in r.items() if "stc" in k or "connector" in k})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/r04t_bench.err
( timeout 1200 python -m pytest tests/test_gpu_parity_full.py::test_configs1_full_depth_end_to_end_fp16_build -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/r04t_pytest_fp16.log 2>&1
grep -E "parity-full|passed|failed|Error|assert" $O/r04t_pytest_fp16.log | cut -c1-220 | grep -v "decode step [0-9]* logits" | tail -14
cp $O/r04_parity.json $O/r04t_parity_fp16.json 2>/dev/null

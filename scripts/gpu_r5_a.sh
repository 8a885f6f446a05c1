#!/bin/bash
# round 5, call A: the fill-the-round GEMM tiles (csrc/k_gemm7.h): bit-identity tests on the GPU, micro-benchmark against the other tile
# shapes, the step with and without them (stage flag 128 = VL2_STAGE_NO_FILL_TILES) alternating on one box, and a first look at the
# power / clock files of the box (scripts/power_trace.py) for VERDICT r04 item 3.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "fill_round or pingpong" > $O/pytest_fill.log 2>&1; echo "pytest rc $?" >> $O/pytest_fill.log
tail -5 $O/pytest_fill.log
timeout 600 python scripts/gemm7_bench.py 3 > $O/gemm7_bench.txt 2>&1; cat $O/gemm7_bench.txt
for rep in 1 2; do for f in 0 128; do
  timeout 600 python bench.py --stage-flags $f --no-cpu-baseline --no-vit-only --steps 8 --warmup 3 2>$O/bench_f${f}_$rep.err | tail -1 > $O/bench_f${f}_$rep.json
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05a/bench_f*_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "encode", j.get("encode_ms"), "prefill", j.get("prefill_ms"), "decode", j.get("decode_ms_per_token"), "fwd_frac", j.get("forward_mfma_frac"))
    for s in j["roofline"]["shapes"]:
        if s["M"] in (1621, 1521):
            print("   ", s["M"], s["N"], s["K"], s.get("avg_launch_us"), s.get("launches"))
PY
ls /sys/class/drm/ > $O/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $O/sysfs_ls.txt 2>&1
timeout 300 python scripts/power_trace.py $O/power_trace_default.json -- python bench.py --no-cpu-baseline --no-vit-only --new-tokens 4 --steps 40 --warmup 3 > $O/power_default.log 2>&1; tail -2 $O/power_default.log

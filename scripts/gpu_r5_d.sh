#!/bin/bash
# round 5, call D (second session; the gpurun_out/ of calls A-C went with the first session's container): the evidence for k_gemm7.h, the
# weave switch and the fp8 matrix-pipe prefill again, in one call: GPU tests of the GEMM family + fp8, micro-benchmark of the fill-the-round
# tiles, the default step against stage flags 128 (no fill tiles) / 256 (weave) alternating, the fp8 prefill line, a rocprofv3 kernel trace
# of the bench command, and the power / clock trace with the persistent GEMM on / off.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
date +%s > $O/t0
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or fill_round or pingpong" -p no:cacheprovider > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/pytest_gemm.log
tail -3 $O/pytest_gemm.log
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s -p no:cacheprovider > $O/pytest_fp8.log 2>&1; echo "pytest rc $?" >> $O/pytest_fp8.log
grep -E "^\[fp8\]|passed|failed|rc |Error|error" $O/pytest_fp8.log | tail -12
timeout 600 python scripts/gemm7_bench.py 2 > $O/gemm7_bench.txt 2>&1; cat $O/gemm7_bench.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_T16.json 2> $O/bench_T16.err
for rep in 1 2 3; do for f in 0 128 256; do
  timeout 600 python bench.py --stage-flags $f --no-cpu-baseline --no-vit-only --steps 8 --warmup 3 2>$O/bench_f${f}_$rep.err | tail -1 > $O/bench_f${f}_$rep.json
done; done
timeout 900 python bench.py --prefill-weights fp8 --decode-weights fp8 --no-cpu-baseline --steps 8 --warmup 3 2>$O/bench_fp8.err | tail -1 > $O/bench_fp8.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
ls /sys/class/drm/ > $O/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $O/sysfs_ls.txt 2>&1
for f in 0 1 0 1; do
  timeout 300 python scripts/power_trace.py $O/power_trace_f${f}_$RANDOM.json -- python bench.py --stage-flags $f --no-cpu-baseline --no-vit-only --new-tokens 1 --steps 200 --warmup 5 > $O/power_f$f.log 2>&1; tail -2 $O/power_f$f.log
done
date +%s > $O/t1
python - <<'PY'
import json, glob, collections
tab = collections.defaultdict(lambda: collections.defaultdict(list))
tot = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05d/bench_*.json")):
    flag = f.split("bench_")[1].rsplit(".", 1)[0]
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(flag, "value", j["value"], "encode", j["encode_ms"], "prefill", j["prefill_ms"], "decode", j["decode_ms_per_token"], "fwd", j["forward_mfma_frac"],
          "roof", j["roofline"]["frac"], {k: v for k, v in j.get("prefill_fp8", {}).items() if k != "arithmetic"})
    fl = flag.split("_")[0]
    for s in j["roofline"]["shapes"]:
        tab[(s["M"], s["N"], s["K"])][fl].append(s["avg_launch_us"])
for k, d in sorted(tab.items(), key=lambda kv: -max(sum(x) for x in kv[1].values())):
    print(k, {fl: [round(x, 1) for x in xs] for fl, xs in sorted(d.items())})
PY

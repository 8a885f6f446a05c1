#!/bin/bash
# round 5, call B: the step with the woven LDS-DMA issue (default) / without (stage flag 256) and with / without the fill-the-round tiles
# (stage flag 128), alternating on one box; per-shape in-pipeline timings from roofline.shapes.  384 = both off = the round-4 kernels.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest rc $?" >> $O/pytest_gemm.log
tail -3 $O/pytest_gemm.log
for rep in 1 2 3; do for f in 0 384 256 128; do
  timeout 600 python bench.py --stage-flags $f --no-cpu-baseline --no-vit-only --steps 8 --warmup 3 2>$O/bench_f${f}_$rep.err | tail -1 > $O/bench_f${f}_$rep.json
done; done
python - <<'PY'
import json, glob, collections
tab = collections.defaultdict(lambda: collections.defaultdict(list))
tot = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05b/bench_f*_*.json")):
    flag = f.split("bench_f")[1].split("_")[0]
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    tot[flag].append((j["encode_ms"], j["prefill_ms"], j["decode_ms_per_token"], j["forward_mfma_frac"]))
    for s in j["roofline"]["shapes"]:
        tab[(s["M"], s["N"], s["K"])][flag].append(s["avg_launch_us"])
for flag, v in sorted(tot.items()):
    print("flags", flag, "(encode, prefill, decode/token, fwd frac):", v)
for k, d in sorted(tab.items(), key=lambda kv: -max(sum(x) for x in kv[1].values())):
    print(k, {fl: [round(x, 1) for x in xs] for fl, xs in sorted(d.items())})
PY

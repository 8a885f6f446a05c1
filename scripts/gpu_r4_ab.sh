#!/bin/bash
# round 4, call AB: per-shape GEMM timings IN THE PIPELINE (bench.py roofline.shapes: HIP events around every launch of the step) with the
# kernel variant forced for every GEMM (--tune gemm=V) against the library's own per-shape choice (V = 0) -- does the chooser, fitted on
# back-to-back micro-benchmarks, still pick the fastest kernel when the operands arrive cold?
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04ab; mkdir -p $O
for rep in 1 2; do for v in 0 1 4 8 12; do
  timeout 600 python bench.py --tune gemm=$v --no-cpu-baseline --no-vit-only --new-tokens 4 --steps 6 --warmup 2 2>$O/bench.err | tail -1 > $O/bench_v${v}_$rep.json
done; done
python - <<'PY'
import json, glob, collections
tab = collections.defaultdict(dict)
tot = {}
for f in sorted(glob.glob("gpurun_out/r04ab/bench_v*_*.json")):
    v = f.split("bench_v")[1].split("_")[0]
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    for s in j["roofline"]["shapes"]:
        k = (s["M"], s["N"], s["K"])
        tab[k].setdefault(v, []).append(s["avg_launch_us"])
    tot.setdefault(v, []).append(round(j["encode_ms"] + j["prefill_ms"], 2))
print("forward ms per variant:", tot)
for k, d in sorted(tab.items(), key=lambda kv: -max(sum(x) for x in kv[1].values())):
    print(k, {v: [round(x, 1) for x in xs] for v, xs in sorted(d.items())})
PY

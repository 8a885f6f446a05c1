#!/bin/bash
# round 5, call J: WEAVE4 -- micro-benchmark on the step's ping-pong shapes, then the step with / without (stage flag 8192), alternating.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05j; mkdir -p $O
timeout 600 python scripts/weave4_bench.py 3 2>&1 | grep -v amdgpu.ids | tee $O/weave4_bench.txt
for rep in 1 2 3; do for f in 0 8192; do
  timeout 600 python bench.py --stage-flags $f --no-cpu-baseline --no-vit-only --steps 8 --warmup 3 2>$O/bench_f${f}_$rep.err | tail -1 > $O/bench_f${f}_$rep.json
done; done
python - <<'PY'
import json, glob, collections
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r05j/bench_f*_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-1500:]); continue
    fl = f.split("bench_f")[1].split("_")[0]
    print(f.split("/")[-1], "value", j["value"], "encode", j["encode_ms"], "prefill", j["prefill_ms"], "decode", j["decode_ms_per_token"], "fwd", j["forward_mfma_frac"], "roof", j["roofline"]["frac"])
    for s in j["roofline"]["shapes"]:
        tab[(s["M"], s["N"], s["K"])][fl].append(s["avg_launch_us"])
for k, d in sorted(tab.items(), key=lambda kv: -max(sum(x) for x in kv[1].values())):
    print(k, {fl: [round(x, 1) for x in xs] for fl, xs in sorted(d.items())})
PY

#!/bin/bash
# round 4, call L: per-shape GEMM timings IN the pipeline (bench.py roofline.shapes), new vs old build on one box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for which in new old new old; do
  case $which in old) cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so;; *) cp /tmp/lib_new.so videollama2_amd/libvl2hip.so;; esac
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04l_bench_${which}.err | tail -1 > $O/r04l_bench_${which}_$RANDOM.json
done
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04l_bench_*.json')):
    j = json.loads(open(f).read())
    print(f.split('/')[-1], {k: j[k] for k in ('encode_ms', 'prefill_ms')}, j['vit_only']['ms'], j['roofline']['frac'])
    print('   ', [(s['M'], s['N'], s['K'], s['avg_launch_us']) for s in j['roofline']['shapes'][:9]])
PY

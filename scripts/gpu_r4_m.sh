#!/bin/bash
# round 4, call M: box classification with per-shape in-pipeline GEMM timings + power / clock samples (rocm-smi every ~0.1 s) during each run
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -E "Max|Level" | tee $O/r04m_smi.txt
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for which in new old new old; do
  case $which in old) cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so;; *) cp /tmp/lib_new.so videollama2_amd/libvl2hip.so;; esac
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; sleep 0.1; done ) > $O/r04m_smi_$which.log 2>&1 &
  SMI=$!
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 --steps 20 2> $O/r04m_bench_${which}.err | tail -1 > $O/r04m_bench_${which}_$RANDOM.json
  kill $SMI
done
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
python - <<'PY'
import json, glob, re
for f in sorted(glob.glob('gpurun_out/r04m_bench_*.json')):
    j = json.loads(open(f).read())
    print(f.split('/')[-1], {k: j[k] for k in ('encode_ms', 'prefill_ms')}, j['vit_only']['ms'], j['roofline']['frac'])
    print('   ', [(s['M'], s['N'], s['K'], s['avg_launch_us']) for s in j['roofline']['shapes'][:9]])
for w in ('new', 'old'):
    pw, ck = [], []
    for ln in open(f'gpurun_out/r04m_smi_{w}.log'):
        m = re.search(r'Power \(W\): ([0-9.]+)', ln); c = re.search(r'sclk clock level: \d+: \((\d+)Mhz', ln)
        if m: pw.append(float(m.group(1)))
        if c: ck.append(int(c.group(1)))
    pw.sort(); ck.sort()
    print(w, 'power samples', len(pw), 'top quartile mean', round(sum(pw[-len(pw)//4:]) / max(1, len(pw[-len(pw)//4:])), 1), 'max', pw[-1:] , 'sclk median', ck[len(ck)//2] if ck else None, 'min', ck[:1])
PY

#!/bin/bash
# round 4, call AA: the row table of the norm-carrying gemm4 GEMMs by LDS-DMA (first MFMA phase waits for slab 0, not for the whole ring fill:
# ADVICE r03) against the pinned register loads -- two builds of the library alternating on one box (bench line + per-shape in-pipeline GEMM
# timings), then the GEMM / stage GPU tests with the new build
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04aa; mkdir -p $O
L=videollama2_amd/libvl2hip.so
cp $L /tmp/new.so
for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then cp videollama2_amd/libvl2hip_tabold.so $L; else cp /tmp/new.so $L; fi
  timeout 600 python bench.py --no-cpu-baseline --no-vit-only --new-tokens 8 --steps 8 --warmup 2 2>$O/bench.err | tail -1 > $O/bench_${v}_$rep.json
  python -c "
import json; j=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
sh={(s['M'],s['N'],s['K']):s['avg_launch_us'] for s in j['roofline']['shapes']}
print('$v', {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, 'gateup', sh.get((1621,28672,4096)), 'qkv', sh.get((1621,6144,4096)), 'vit_qkv', sh.get((9232,3072,1024)), 'roof', j['roofline']['frac'])" | tee -a $O/ab.txt
done; done
cp /tmp/new.so $L
( timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py -m gpu -q -p no:cacheprovider -k "gemm or stage" 2>&1 ) > $O/pytest_gemm.log 2>&1; tail -3 $O/pytest_gemm.log | cut -c1-300

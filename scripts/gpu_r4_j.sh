#!/bin/bash
# round 4, call J: CU census (does a grid of one 139 KB-LDS workgroup per CU start all at once on this box?) + box classification
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
timeout 60 scripts/ubench/cu_census 139280 | tee $O/r04j_census.txt
timeout 60 scripts/ubench/cu_census 131072 | tee -a $O/r04j_census.txt
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for which in new old new old; do
  if [ $which = old ]; then cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so; else cp /tmp/lib_new.so videollama2_amd/libvl2hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04j_bench_${which}.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
done | tee $O/r04j_box_class.txt
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
timeout 60 scripts/ubench/cu_census 139280 | tee -a $O/r04j_census.txt

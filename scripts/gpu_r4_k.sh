#!/bin/bash
# round 4, call K: CU census + three builds alternating (new = persistent, first tile static; fd = first tile from the counter too; old = none)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
timeout 60 scripts/ubench/cu_census 139280 | grep "G 256" | tee $O/r04k_census.txt
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for i in 1 2; do
for which in new old fd; do
  case $which in old) cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so;; fd) cp scripts/ubench/libvl2hip_firstdyn.so videollama2_amd/libvl2hip.so;; *) cp /tmp/lib_new.so videollama2_amd/libvl2hip.so;; esac
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04k_bench_${which}.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
done; done | tee $O/r04k_box_class.txt
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
timeout 60 scripts/ubench/cu_census 139280 | grep "G 256" | tee -a $O/r04k_census.txt
export LAB_SHAPES=vit_qkv_ln,vit_fc1_ln
timeout 300 scripts/ubench/gemm_lab 2 8,9,60,70,80 2>/dev/null | tee $O/r04k_lab.txt

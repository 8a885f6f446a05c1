#!/bin/bash
# round 4, call H: is the persistent GEMM's pipeline loss a cold-cache effect?  lab warm vs LAB_COLD (600 MB memset before every timed
# launch) on the same box, + one new / old bench pair to classify the box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for which in new old; do
  if [ $which = old ]; then cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so; else cp /tmp/lib_new.so videollama2_amd/libvl2hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04h_bench_${which}.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
done | tee $O/r04h_box_class.txt
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
export LAB_SHAPES=vit_qkv_ln,vit_fc1_ln,stc_s1_b1,llm_gateup_rms
echo warm; timeout 300 scripts/ubench/gemm_lab 2 8,9,17,60,70,61,71 2>/dev/null | tee $O/r04h_lab_warm.txt
echo cold; LAB_COLD=1 timeout 300 scripts/ubench/gemm_lab 2 8,9,17,60,70,61,71 2>/dev/null | tee $O/r04h_lab_cold.txt
echo cold2; LAB_COLD=2 timeout 300 scripts/ubench/gemm_lab 2 8,9,17,60,70,61,71 2>/dev/null | tee $O/r04h_lab_cold2.txt

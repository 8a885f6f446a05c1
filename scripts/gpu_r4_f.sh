#!/bin/bash
# round 4, call F: why does the persistent GEMM win per kernel and lose in the pipeline?  (1) kernel_bench with 400 iterations per variant
# (power steady state instead of 20-launch bursts); (2) the pipeline with the persistent form on the K < 2048 calls only / K >= 2048 only
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
timeout 600 python scripts/kernel_bench.py --frames 16 --iters 400 --quick --only vit_qkv,vit_fc1,stc_s1_b1,stc_s1_conv,llm_gateup > $O/r04f_kernel_bench_sustained.txt 2> $O/r04f_kb.err
cat $O/r04f_kernel_bench_sustained.txt
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for i in 1 2; do
  for which in new old p1 p2; do
    case $which in old) cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so;; p1) cp scripts/ubench/libvl2hip_persist1.so videollama2_amd/libvl2hip.so;;
      p2) cp scripts/ubench/libvl2hip_persist2.so videollama2_amd/libvl2hip.so;; *) cp /tmp/lib_new.so videollama2_amd/libvl2hip.so;; esac
    timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04f_bench_${which}_$i.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', $i, {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
  done
done | tee $O/r04f_persist_ab.txt
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so

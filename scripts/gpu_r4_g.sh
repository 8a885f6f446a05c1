#!/bin/bash
# round 4, call G: persistent GEMM with DYNAMIC tile hand-out -- lab (v60 / v61 static walk, v70 / v71 dynamic; == against gemm4), then the
# pipeline with three builds alternating on one box (new = automatic choice incl. the persistent form, p1 = persistent on K < 2048 calls only,
# old = compiled out), then the GPU tests of the GEMM / stage paths
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
export LAB_SHAPES=vit_qkv,vit_fc1,stc_s1,llm_gateup,sq_8192x4096
timeout 300 scripts/ubench/gemm_lab 3 8,9,17,60,70,61,71 > $O/r04g_gemm_lab.txt 2> $O/r04g_gemm_lab.err
echo "lab rc=$?" >> $O/r04g_gemm_lab.txt; cat $O/r04g_gemm_lab.txt; grep -v hash $O/r04g_gemm_lab.err | tail
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for i in 1 2; do
  for which in new old p1; do
    case $which in old) cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so;; p1) cp scripts/ubench/libvl2hip_persist1.so videollama2_amd/libvl2hip.so;;
      *) cp /tmp/lib_new.so videollama2_amd/libvl2hip.so;; esac
    timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04g_bench_${which}_$i.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', $i, {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
  done
done | tee $O/r04g_persist_ab.txt
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
( timeout 1500 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/r04g_pytest.log 2>&1
tail -3 $O/r04g_pytest.log

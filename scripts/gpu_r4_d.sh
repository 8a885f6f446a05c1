#!/bin/bash
# round 4, call D: the persistent GEMM in the PIPELINE -- two builds of the library (HEAD, and HEAD with choose_gemm6 compiled out)
# alternating on one box; then the stage / parity GPU tests at HEAD
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out
mkdir -p $O
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for i in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so; else cp /tmp/lib_new.so videollama2_amd/libvl2hip.so; fi
    timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04d_bench_${which}_$i.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', $i, {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
  done
done | tee $O/r04d_persist_ab.txt
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
( timeout 1500 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/r04d_pytest.log 2>&1
tail -3 $O/r04d_pytest.log

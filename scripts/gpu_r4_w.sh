#!/bin/bash
# round 4, call W: attention softmax on packed fp32 (v_pk_fma_f32 / v_pk_add_f32) against the previous kernel -- two builds of the library
# alternating on one box (micro-benchmark + the bench line), then the attention GPU tests with the new build
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04w; mkdir -p $O
L=videollama2_amd/libvl2hip.so
cp $L /tmp/new.so
for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then cp videollama2_amd/libvl2hip_attnold.so $L; else cp /tmp/new.so $L; fi
  echo "$v $(timeout 300 python scripts/attn_ab.py 2>/dev/null | tail -1)" | tee -a $O/attn_ab.txt
  timeout 600 python bench.py --no-cpu-baseline --no-vit-only --steps 5 --warmup 2 2>$O/bench.err | tail -1 | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k: j[k] for k in ('encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac')})" | tee -a $O/bench_ab.txt
done; done
cp /tmp/new.so $L
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "attn" 2>&1 ) > $O/pytest_attn.log 2>&1; tail -3 $O/pytest_attn.log | cut -c1-300

#!/bin/bash
# Round 3, GPU call E: A/B of the in-kernel statistics reduction against the finalize launches (two builds of the library,
# alternating runs on one box), + the stage / bit-identity GPU tests
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r03t}
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/pytest_stage.log 2>&1
grep -E "passed|failed|error" $O/pytest_stage.log | tail -3
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for i in 1 2 3; do
  for which in new fin; do
    if [ $which = fin ]; then cp $2 videollama2_amd/libvl2hip.so; else cp /tmp/lib_new.so videollama2_amd/libvl2hip.so; fi
    timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/bench_$which_$i.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', $i, {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'])"
  done
done
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so

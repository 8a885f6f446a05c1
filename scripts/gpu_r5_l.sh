#!/bin/bash
# round 5, call L: gemm9 (the 256 x 256 ping-pong tile on v_mfma_f32_16x16x32_bf16, opt-in): tests, micro-benchmark interleaved with the family and the vendor,
# the T = 16 line with and without VL2_STAGE_MFMA16.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py -x -q -k "mfma16 or stage_calls_equal" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log | cut -c1-300
timeout 600 python scripts/mfma16_bench.py 3 > $O/mfma16_bench.txt 2>&1; echo "bench rc $?"
cat $O/mfma16_bench.txt | cut -c1-400
i=0
for fl in 0 32768 0 32768; do
  i=$((i+1))
  timeout 600 python bench.py --no-cpu-baseline --no-vit-only --stage-flags $fl > $O/bench_T16_flags${fl}_$i.json 2> $O/bench_T16_flags${fl}_$i.err
  python -c "
import json; j=json.loads(open('$O/bench_T16_flags${fl}_$i.json').read().strip().splitlines()[-1]); print($fl, {k: j.get(k) for k in ('value','ms_per_step','encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac')})"
done

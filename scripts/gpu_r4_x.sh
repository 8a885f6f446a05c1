#!/bin/bash
# round 4, call X: fp8 decode weights (the fp8 half) on hardware -- the GPU tests of tests/test_gpu_fp8.py (quantiser bit for
# bit against the oracle, GEMV at the decoder's shapes, graph == eager, kernels vs the 16-bit kernels on the dequantised weights), the GEMV
# micro-benchmark, the bench line with the `decode_fp8` key (twice), the decode-step GPU tests that share code with it
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04x; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_fp8.py -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_fp8.log 2>&1
grep -E "^\[fp8\]|passed|failed|Error|assert" $O/pytest_fp8.log | cut -c1-260 | tail -30
cp gpurun_out/r04_fp8_parity.json $O/ 2>/dev/null
timeout 300 python scripts/fp8_bench.py 2>/dev/null | tee $O/fp8_bench.txt
for i in 1 2; do
  timeout 600 python bench.py --decode-weights fp8 --no-cpu-baseline --no-vit-only 2>$O/bench.err | tail -1 > $O/bench_fp8_$i.json
  python -c "
import json; j=json.loads(open('$O/bench_fp8_$i.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac')}, j.get('decode_fp8'))" 2>&1 | cut -c1-600
done
tail -3 $O/bench.err
( timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_stage_abi.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/pytest_stage.log 2>&1; tail -2 $O/pytest_stage.log

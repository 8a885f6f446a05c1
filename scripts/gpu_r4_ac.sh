#!/bin/bash
# round 4, call AC: configs[1] end to end at full depth again, now with the fp8-weights decode rows (format error at 32 layers) appended
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04ac; mkdir -p $O
( timeout 1500 python -m pytest "tests/test_gpu_parity_full.py::test_configs1_full_depth_end_to_end" -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_parity.log 2>&1
grep -E "fp8-weights|passed|failed|Error|assert" $O/pytest_parity.log | cut -c1-300 | tail -12
cp gpurun_out/r04_parity.json $O/ 2>/dev/null

#!/bin/bash
# round 4, call AE: the two full-depth configs[1] cases back to back (the second reuses the first one's seeded weights) + free host memory
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04ae; mkdir -p $O
free -g | head -2
( time timeout 1500 python -m pytest tests/test_gpu_parity_full.py::test_configs1_full_depth_end_to_end tests/test_gpu_parity_full.py::test_configs1_full_depth_end_to_end_fp16_build -m gpu -q -p no:cacheprovider --durations=4 2>&1 ) > $O/pytest_parity.log 2>&1
grep -E "s call|passed|failed|^real|Error" $O/pytest_parity.log | cut -c1-200 | tail -8

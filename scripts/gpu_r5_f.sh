#!/bin/bash
# round 5, call F: the hardened parity rows (full-depth end to end at T = 8 and T = 32, the outlier fixture at 12 + 8 layers in
# both element types with decidable steps, the fp8 decode against the fp32 oracle on the DEQUANTISED weights with a bar) -> r05_parity.json;
# the phase-resolved power / clock trace again, now on the card that is actually this container's GPU; stage-level ticket check.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
date +%s > $O/t0
timeout 600 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_ops.py -x -q -k "stage_calls or producer_side" -p no:cacheprovider > $O/pytest_stage.log 2>&1; echo "pytest rc $?" >> $O/pytest_stage.log
tail -3 $O/pytest_stage.log
timeout 600 python scripts/phase_power_ab.py $O/phase_power_ab.json --flags 0,1 --seconds 2.5 --reps 2 > $O/phase_power_ab.txt 2>&1; grep -v amdgpu.ids $O/phase_power_ab.txt | tail -14
timeout 2400 python -m pytest tests/test_gpu_parity_full.py -x -q -s -p no:cacheprovider --durations=12 > $O/pytest_parity.log 2>&1; echo "pytest rc $?" >> $O/pytest_parity.log
grep -E "^\[parity-full\]|passed|failed|rc |Error|error|assert" $O/pytest_parity.log | cut -c1-220 | tail -150
cp gpurun_out/r05_parity.json $O/ 2>/dev/null
date +%s > $O/t1

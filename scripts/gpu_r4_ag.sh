#!/bin/bash
# round 4, call AG: the full GPU suite at the last commit of the round (after the test-infrastructure changes: parallel seeded weights, the
# fp16 full-depth case reusing the bf16 case's weights) + smoke
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04ag; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=6 2>&1 ) > $O/pytest_gpu.log 2>&1
grep -E "s call|passed|failed|^real|Error" $O/pytest_gpu.log | cut -c1-200 | tail -12

#!/bin/bash
# Round 3: parity evidence -- tests/test_gpu_parity_full.py (configs[1] per stage + END TO END at full depth) and the token tests
# whose escape hatches were replaced by the strict margin < 2 x max|dlogit| rule
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r03p}
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -p no:cacheprovider -s -x 2>&1 ) > $O/pytest_parity_full.log 2>&1
cp gpurun_out/r03_parity.json $O/ 2>/dev/null
( time timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_v21.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/pytest_tokens.log 2>&1
grep -E "^\[parity|passed|failed|error|FAILED|ERROR|real" $O/pytest_parity_full.log | tail -60
tail -5 $O/pytest_tokens.log

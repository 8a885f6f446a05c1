#!/bin/bash
# round 5, call O: the GEMM-family and stage tests on the library as rebuilt with gemm9's last lab forms (variants 25 / 26)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05o; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stage_abi.py -x -q -k "gemm or stage_calls" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log | cut -c1-300

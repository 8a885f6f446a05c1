#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attn" 2>&1 | tail -2
timeout 600 python scripts/kernel_bench.py --quick 2>&1 | grep "^attn" 
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/apmc2 -o pmc -- python $GRAFT_REPO_ROOT/scripts/attn_pmc.py > /dev/null 2>&1; echo "pmc exit $?"

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gemv_fetch -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemv_traffic_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/gemv_fetch.log 2>&1; echo "gemv pmc exit $?"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/apmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/attn_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/apmc_$tag.log 2>&1
  echo "attn pmc $tag exit $?"
done

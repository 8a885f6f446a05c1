#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gpmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/gemm_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/gpmc_$tag.log 2>&1
  echo "pmc $tag exit $?"
done

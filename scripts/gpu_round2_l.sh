#!/bin/bash
# Round 2, GPU call L: PMC counters of the two attention kernels of the head of the round (two separate --pmc passes, kernel-trace only).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02l}
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/apmc_$i -o pmc -- python $R/scripts/attn_pmc.py > $R/$O/apmc_$i.log 2>&1 )
  F=$(find $O/apmc_$i -name "*counter_collection.csv" | head -1); cp "$F" $O/attn_pmc_pass$i.csv; rm -rf $O/apmc_$i
done
echo done

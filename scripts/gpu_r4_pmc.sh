#!/bin/bash
# round 4: SQ / GRBM counters of the GEMM kernels HEAD dispatches (three separate --pmc passes, kernel-trace only) -> one CSV
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=gpurun_out/r04pmc; mkdir -p $O
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/pass$i -o pmc -- python $R/scripts/gemm_pmc.py > $R/$O/pass$i.log 2>&1 ); echo "pass $i exit $?"
done
python scripts/gemm_pmc_post.py $O/pass1 $O/pass2 > $O/r04_gemm_pmc_counters.csv 2> $O/post.err; tail -2 $O/post.err
rm -rf $O/pass1 $O/pass2
cut -d, -f1,2,3,12,20- $O/r04_gemm_pmc_counters.csv | head -12

#!/bin/bash
# round 4, call E: rocprofv3 kernel traces of the bench with the two library builds (persistent GEMM on / compiled out), same box
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=${GRAFT_REPO_ROOT:-$(pwd)}
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for which in new old; do
  if [ $which = old ]; then cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so; else cp /tmp/lib_new.so videollama2_amd/libvl2hip.so; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$which -o bench -- python $R/bench.py --steps 3 --warmup 1 --new-tokens 2 --no-cpu-baseline --no-vit-only > $R/$O/r04e_trace_$which.log 2>&1 )
  find $O/trace_$which -name "*kernel_stats.csv" -exec cp {} $O/r04e_kernel_stats_$which.csv \;
  find $O/trace_$which -name "*kernel_trace.csv" -exec cp {} $O/r04e_kernel_trace_$which.csv \;
  rm -rf $O/trace_$which
done
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
head -12 $O/r04e_kernel_stats_new.csv | cut -c1-160; echo; head -12 $O/r04e_kernel_stats_old.csv | cut -c1-160

#!/bin/bash
# round 4, call I: classify the box (new / old bench pair), then kernel traces of the ViT-heavy bench with both builds on the SAME box
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; R=${GRAFT_REPO_ROOT:-$(pwd)}
cp videollama2_amd/libvl2hip.so /tmp/lib_new.so
for which in new old new old; do
  if [ $which = old ]; then cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so; else cp /tmp/lib_new.so videollama2_amd/libvl2hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 4 2> $O/r04i_bench_${which}.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', {k: j[k] for k in ('encode_ms','prefill_ms','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
done | tee $O/r04i_box_class.txt
for which in new old; do
  if [ $which = old ]; then cp scripts/ubench/libvl2hip_nopersist.so videollama2_amd/libvl2hip.so; else cp /tmp/lib_new.so videollama2_amd/libvl2hip.so; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$which -o bench -- python $R/bench.py --steps 3 --warmup 1 --new-tokens 2 --no-cpu-baseline > $R/$O/r04i_trace_$which.log 2>&1 )
  find $O/trace_$which -name "*kernel_stats.csv" -exec cp {} $O/r04i_kernel_stats_$which.csv \;
  find $O/trace_$which -name "*kernel_trace.csv" -exec cp {} $O/r04i_kernel_trace_$which.csv \;
  rm -rf $O/trace_$which
done
cp /tmp/lib_new.so videollama2_amd/libvl2hip.so
rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | head -30 > $O/r04i_smi.txt; rocm-smi --showmaxpower --showmemuse 2>/dev/null | head -20 >> $O/r04i_smi.txt

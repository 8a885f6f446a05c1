#!/bin/bash
# round 5: one more box for the question "does the persistent GEMM win in the tower on EVERY box?" (round 4 recorded a 1-ms loss on 3 of 11 boxes,
# evidence lost): the CLIP tower alone, stage flags 0 / 1 alternating, 3 repetitions of 2 s, with the power / clock samples.  $1 = tag of the call.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05vit; mkdir -p $O
timeout 300 python scripts/phase_power_ab.py $O/vit_ab_$1.json --flags 0,1 --seconds 2.0 --reps 3 --phases vit 2>&1 | grep -v amdgpu.ids | tee $O/vit_ab_$1.txt
rocm-smi --showproductname --showuniqueid 2>/dev/null | grep -iE "unique|series|GPU\[" | head -4 >> $O/vit_ab_$1.txt

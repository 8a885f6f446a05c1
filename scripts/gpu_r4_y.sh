#!/bin/bash
# round 4, call Y: what bounds vl2_gemv_fp8 (4.3 TB/s of fp8 bytes against 6.4 TB/s of 16-bit bytes)?  The same kernel with the fp8 -> element
# conversion compiled out (-DVL2_FP8_LAB_RAW: raw dwords meet x; wrong numbers, right memory traffic) beside the product build, alternating
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04y; mkdir -p $O
L=videollama2_amd/libvl2hip.so
cp $L /tmp/new.so
for rep in 1 2; do for v in product raw; do
  if [ $v = raw ]; then cp videollama2_amd/libvl2hip_fp8raw.so $L; else cp /tmp/new.so $L; fi
  echo "== $v" | tee -a $O/fp8_lab.txt
  timeout 300 python scripts/fp8_bench.py 2>/dev/null | grep -v JSON | tee -a $O/fp8_lab.txt
done; done
cp /tmp/new.so $L

#!/bin/bash
# Round 2, GPU call H: attention D = 64 at 4 waves per SIMD (128 VGPRs, 60 B/lane of scratch) against the shipped 3 waves per SIMD
# (142 VGPRs, no scratch): two builds of the library, alternating processes on one box.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r02h}
mkdir -p $O
cp videollama2_amd/libvl2hip.so /tmp/lib_a.so
for r in 1 2; do
  cp /tmp/lib_a.so videollama2_amd/libvl2hip.so; timeout 200 python scripts/attn_bench2.py 2>/dev/null | head -3 > $O/attn_occ3_r$r.jsonl
  cp gpurun_aux/libvl2hip_occ4.so videollama2_amd/libvl2hip.so; timeout 200 python scripts/attn_bench2.py 2>/dev/null | head -3 > $O/attn_occ4_r$r.jsonl
done
cp /tmp/lib_a.so videollama2_amd/libvl2hip.so
echo done

#!/bin/bash
# row-split GEMMs as ONE mixed launch (big tiles + 128x128 tail tiles) vs two launches (VL2_GEMM_NO_MIX=1): bit-identity tests, bench A/B on one box
line() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['encode_ms'], j['prefill_ms'], j['decode_ms_per_token'], j['forward_mfma_frac'], j['roofline']['frac'], j['roofline']['dominant']['avg_launch_us'])"; }
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stages.py tests/test_gpu_stage_abi.py -m gpu -q -p no:cacheprovider -k "gemm or shard or stage or connector or tower" 2>&1 | tail -1
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-vit-only 2>/dev/null | line mix
  VL2_GEMM_NO_MIX=1 timeout 300 python bench.py --no-cpu-baseline --no-vit-only 2>/dev/null | line two-launch
done
timeout 300 python bench.py --frames 8 --no-cpu-baseline --no-vit-only 2>/dev/null | line mix_T8
VL2_GEMM_NO_MIX=1 timeout 300 python bench.py --frames 8 --no-cpu-baseline --no-vit-only 2>/dev/null | line two-launch_T8
timeout 300 python bench.py --frames 32 --no-cpu-baseline --no-vit-only 2>/dev/null | line mix_T32
VL2_GEMM_NO_MIX=1 timeout 300 python bench.py --frames 32 --no-cpu-baseline --no-vit-only 2>/dev/null | line two-launch_T32

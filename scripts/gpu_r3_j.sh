#!/bin/bash
# ViT: statistics reduced inside the consuming GEMMs (behind their first LDS-DMA slabs) vs the row_norm_finalize launches (VL2_VIT_FINALIZE=1), same box, alternating
line() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['encode_ms'], j['prefill_ms'], j['decode_ms_per_token'], j['forward_mfma_frac'], j['roofline']['frac'])"; }
timeout 600 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-vit-only 2>/dev/null | line in-gemm
  VL2_VIT_FINALIZE=1 timeout 300 python bench.py --no-cpu-baseline --no-vit-only 2>/dev/null | line finalize
done

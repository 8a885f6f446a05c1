#!/bin/bash
# same-box A/B: decode step with the two-kernel attention (default) vs the fused attention + combine (VL2_DECODE_FUSED_ATTN=1)
line() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['encode_ms'], j['prefill_ms'], j['decode_ms_per_token'], j['ms_per_step'])"; }
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line unfused
  VL2_DECODE_FUSED_ATTN=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line fused
done
VL2_DECODE_FUSED_ATTN=1 timeout 300 python -m pytest tests/test_gpu_v21.py tests/test_gpu_stages.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1

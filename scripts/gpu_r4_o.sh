#!/bin/bash
# round 4, call O: decode tail engine -- correctness (stage == per-operator, graph == eager, batched == single on the device) and the decode
# rate with / without it (stage flag 16 = three GEMV launches), twice alternating; attention with the class token peeled
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_v21.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/r04o_pytest.log 2>&1; tail -4 $O/r04o_pytest.log
for f in 0 16 0 16; do
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 64 --stage-flags $f --no-vit-only 2>> $O/r04o_bench.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage-flags $f', {k: j[k] for k in ('encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac')})"
done | tee $O/r04o_decode_ab.txt
timeout 600 python - <<'PY' 2>/dev/null | tee $O/r04o_attn.txt
import torch, sys
sys.path.insert(0, '.')
from videollama2_amd import ops
from scripts.kernel_bench import timeit, rnd
B, H, Nn, D = 16, 16, 577, 64
qkv = rnd(B * Nn, 3 * H * D)
o = torch.empty(B * Nn, H * D, dtype=torch.bfloat16, device='cuda')
st = (Nn * 3 * H * D, D, 3 * H * D)
for rep in range(3):
    for var, name in ((3, 'plain tiling (variant 3)'), (0, 'class token peeled (auto)')):
        ops.set_attn_kv_groups(var)
        us = timeit(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (Nn * H * D, D, H * D), B, H, Nn, Nn, 1, D ** -0.5, False, 0, D), iters=100)
        print(f'attn_vit T=16 {name}: {us:.1f} us')
PY
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -x -k "attn" 2>&1 ) | tail -2

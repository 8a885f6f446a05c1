#!/bin/bash
# round 4, call N: full GPU suite at HEAD (descriptor flags, class-token peel), attention micro-benchmark, bench with / without the persistent GEMM
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 ) > $O/r04n_pytest_gpu.log 2>&1; tail -4 $O/r04n_pytest_gpu.log
timeout 300 python scripts/kernel_bench.py --quick 2>/dev/null | grep -E "attn_|gemv_" | tee $O/r04n_kernel_bench_attn.txt
timeout 600 python - <<'PY' 2>/dev/null | tee -a $O/r04n_kernel_bench_attn.txt
import torch, sys
sys.path.insert(0, '.')
from videollama2_amd import ops
from scripts.kernel_bench import timeit, rnd
B, H, Nn, D = 16, 16, 577, 64
qkv = rnd(B * Nn, 3 * H * D)
o = torch.empty(B * Nn, H * D, dtype=torch.bfloat16, device='cuda')
st = (Nn * 3 * H * D, D, 3 * H * D)
for rep in range(2):
    for var, name in ((3, 'plain tiling (variant 3)'), (0, 'class token peeled (auto)')):
        ops.set_attn_kv_groups(var)
        us = timeit(lambda: ops.attn_fwd(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], o, st, st, st, (Nn * H * D, D, H * D), B, H, Nn, Nn, 1, D ** -0.5, False, 0, D), iters=50)
        print(f'attn_vit T=16 {name}: {us:.1f} us')
ops.set_attn_kv_groups(0)
PY
for f in 0 1 0 1; do
  timeout 600 python bench.py --no-cpu-baseline --new-tokens 8 --stage-flags $f 2>> $O/r04n_bench.err | python -c "
import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage-flags $f', {k: j[k] for k in ('encode_ms','prefill_ms','decode_ms_per_token','forward_mfma_frac')}, j['vit_only']['ms'], j['roofline']['frac'])"
done | tee $O/r04n_bench_flags.txt

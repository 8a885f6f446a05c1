#!/bin/bash
# round 4, call AF: fp8 decode weights on the other family (VideoLLaMA2.1: Qwen2 decoder with q/k/v bias, K = 3584 / 18944)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04af; mkdir -p $O
timeout 600 python bench.py --model v21 --decode-weights fp8 --no-cpu-baseline --no-vit-only 2>$O/bench.err | tail -1 > $O/bench_v21_fp8.json
python -c "
import json; j=json.loads(open('$O/bench_v21_fp8.json').read().strip().splitlines()[-1]); print({k: j[k] for k in ('encode_ms','prefill_ms','decode_ms_per_token','decode_hbm_frac')}, {k: v for k, v in (j.get('decode_fp8') or {}).items() if k != 'what'})"
tail -2 $O/bench.err

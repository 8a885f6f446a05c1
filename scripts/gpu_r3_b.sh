#!/bin/bash
# Round 3, GPU call B: new GPU tests (TP real shards), the bench line with the round-3 input recipe (uint8 frames), and the
# multi-rank bench control flow with the ranks sharing the one GPU over the gloo debug backend (both cuts + ViT-only keys)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r03q}
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_tp.py -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_tp.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench_T16.json 2> $O/bench_T16.err
timeout 600 python bench.py --no-cpu-baseline --bf16-frames > $O/bench_T16_bf16frames.json 2> $O/bench_T16_bf16frames.err
for N in 2 4; do
  VL2_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 2 --warmup 1 --new-tokens 2 --no-cpu-baseline > $O/bench_gloo_$N.json 2> $O/bench_gloo_$N.err
done
grep -E "tp-local|passed|failed|Error" $O/pytest_tp.log | tail; for f in $O/bench_T16.json $O/bench_T16_bf16frames.json $O/bench_gloo_2.json $O/bench_gloo_4.json; do python - <<PY
import json
try:
    j = json.loads(open("$f").read().strip().splitlines()[-1])
    print("$f", {k: j.get(k) for k in ("value", "encode_ms", "prefill_ms", "decode_ms_per_token", "forward_mfma_frac", "vit_only", "cut", "north_star_cut", "sharded_connector_cut", "frames_input")}, j["roofline"].get("dominant"))
except Exception as e:
    print("$f", "ERR", e)
PY
done

#!/bin/bash
# round 5, call G: do_sample (csrc/k_sample.h) on the device; the deep outlier fixture with stationary massive channels in both element types;
# the default step with the persistent GEMM in the tower (new default) against VL2_STAGE_VIT_NO_PERSISTENT (4096), alternating; bench.py --gpus 2 / 4 over
# gloo on the one GPU (control flow + the new per-rank keys; timings meaningless).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_stage_abi.py tests/test_gpu_api.py -x -q -s -p no:cacheprovider > $O/pytest_sampling.log 2>&1; echo "pytest rc $?" >> $O/pytest_sampling.log
grep -E "^\[sampling\]|passed|failed|rc |Error|error" $O/pytest_sampling.log | cut -c1-300 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity_full.py -x -q -s -p no:cacheprovider -k "outlier" > $O/pytest_outliers.log 2>&1; echo "pytest rc $?" >> $O/pytest_outliers.log
grep -E "passed|failed|rc |Error|assert|decidable" $O/pytest_outliers.log | cut -c1-250 | tail -8
grep -E "^\[parity-full\] outliers 12\+8 (bf16|fp16) e2e (prefill|decode step (1|8|16|24|31) )" $O/pytest_outliers.log | cut -c1-200
cp gpurun_out/r05_parity.json $O/r05_parity_outliers.json 2>/dev/null
for rep in 1 2 3; do for f in 0 4096; do
  timeout 600 python bench.py --stage-flags $f --no-cpu-baseline --steps 8 --warmup 3 2>$O/bench_f${f}_$rep.err | tail -1 > $O/bench_f${f}_$rep.json
done; done
export VL2_DIST_BACKEND=gloo
for N in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) \
      bench.py --gpus $N --steps 2 --warmup 1 --new-tokens 8 > $O/bench_gloo_$N.json 2> $O/bench_gloo_$N.err
  echo "gloo N=$N exit $?"
done
unset VL2_DIST_BACKEND
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05g/bench_f*_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-1500:]); continue
    print(f.split("/")[-1], "value", j["value"], "encode", j["encode_ms"], "prefill", j["prefill_ms"], "decode", j["decode_ms_per_token"], "fwd", j["forward_mfma_frac"], "roof", j["roofline"]["frac"], "vit_only", (j.get("vit_only") or {}).get("ms"))
for n in (2, 4):
    try:
        j = json.loads(open(f"gpurun_out/r05g/bench_gloo_{n}.json").read().strip().splitlines()[-1])
        print("gloo", n, "ranks_seen", j.get("ranks_seen"), "equal", j.get("sharded_encoder_equals_unsharded"), "per_rank", json.dumps(j.get("per_rank"))[:600])
    except Exception as e:
        print("gloo", n, "ERR", e, open(f"gpurun_out/r05g/bench_gloo_{n}.err").read()[-1500:])
PY

#!/bin/bash
# round 5, call C: the fp8 matrix-pipe prefill (VL2_GEMM_FP8 / VL2_STAGE_PREFILL_FP8): GPU tests against oracle/fp8_oracle.py, the bench line
# with the `prefill_fp8` key, and the default step again (weave off = round-4 kernels + fill tiles).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fp8.py -x -q -s > $O/pytest_fp8.log 2>&1; echo "pytest rc $?" >> $O/pytest_fp8.log
grep -E "^\[fp8\]|passed|failed|rc |Error|error" $O/pytest_fp8.log | tail -20
timeout 900 python bench.py --prefill-weights fp8 --decode-weights fp8 --no-cpu-baseline --steps 8 --warmup 3 2>$O/bench_fp8.err | tail -1 > $O/bench_fp8.json
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r05c/bench_fp8.json").read().strip().splitlines()[-1])
    print("encode", j["encode_ms"], "prefill", j["prefill_ms"], "decode", j["decode_ms_per_token"], "fwd", j["forward_mfma_frac"])
    print("prefill_fp8", {k: v for k, v in j.get("prefill_fp8", {}).items() if k != "arithmetic"})
    print("decode_fp8", {k: v for k, v in j.get("decode_fp8", {}).items() if k != "what"})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r05c/bench_fp8.err").read()[-3000:])
PY

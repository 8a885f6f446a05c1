#!/bin/bash
# round 5, call E: the producer-side finalize (k_gemm.h gemm_rows_ticket; stage flag 1024 = VL2_STAGE_NO_TICKET keeps the row_norm_finalize
# launches): GPU tests (operator + stage == per-operator), pair micro-benchmark, the step with / without alternating on one box; the
# phase-resolved power / clock trace of the tower, connector and prefill with the persistent GEMM on / off.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "producer_side or norm_carrying or fill_round" -p no:cacheprovider > $O/pytest_ticket.log 2>&1; echo "pytest rc $?" >> $O/pytest_ticket.log
tail -4 $O/pytest_ticket.log
timeout 900 python -m pytest tests/test_gpu_stage_abi.py tests/test_gpu_stages.py tests/test_gpu_fp8.py -x -q -p no:cacheprovider > $O/pytest_stage.log 2>&1; echo "pytest rc $?" >> $O/pytest_stage.log
tail -4 $O/pytest_stage.log
timeout 600 python scripts/ticket_bench.py 3 > $O/ticket_bench.txt 2>&1; cat $O/ticket_bench.txt
for rep in 1 2 3; do for f in 0 1024; do
  timeout 600 python bench.py --stage-flags $f --no-cpu-baseline --no-vit-only --steps 8 --warmup 3 2>$O/bench_f${f}_$rep.err | tail -1 > $O/bench_f${f}_$rep.json
done; done
timeout 600 python scripts/phase_power_ab.py $O/phase_power_ab.json --flags 0,1 --seconds 2.5 --reps 2 > $O/phase_power_ab.txt 2>&1; cat $O/phase_power_ab.txt | tail -14
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05e/bench_f*_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-1500:]); continue
    print(f.split("/")[-1], "value", j["value"], "encode", j["encode_ms"], "prefill", j["prefill_ms"], "decode", j["decode_ms_per_token"], "fwd", j["forward_mfma_frac"], "roof", j["roofline"]["frac"])
PY

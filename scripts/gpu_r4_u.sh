#!/bin/bash
# round 4, call U (re-entry after the container was re-created: the earlier calls' gpurun_out/ is gone): evidence at HEAD in one call --
# smoke, the full GPU suite, the default bench line, its rocprofv3 kernel trace, the connector's direct kernels, per-shape GEMM timings,
# then the round's A/B switches in the PIPELINE, alternating on this one box: stage flags 0 (default) / 1 (persistent GEMM) /
# 32 (five-launch SE chain of rounds 1-3) / 16 (decode tail engine), and the fp16 build's bench line.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04u; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/pytest_gpu_full.log 2>&1
grep -E "^\[parity|^\[rccl|^\[tp-local|passed|failed|error|FAILED|ERROR|^real" $O/pytest_gpu_full.log | cut -c1-400 | tail -150 > $O/pytest_gpu.log
grep -E "passed|failed|^real" $O/pytest_gpu.log | tail -3
cp gpurun_out/r04_parity.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vit-only > $R/$O/trace_bench.log 2>&1 )
rm -f $O/trace/bench_kernel_trace.csv
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench.csv \;
rm -rf $O/trace
timeout 300 python scripts/stc_bench.py 2>&1 | grep -v amdgpu.ids > $O/stc_bench.txt
timeout 600 python scripts/kernel_bench.py --quick 2>/dev/null > $O/kernel_bench_T16.txt
for i in 1 2; do for fl in 0 1 32 16; do
  timeout 600 python bench.py --stage-flags $fl --no-cpu-baseline --no-vit-only --steps 5 --warmup 2 2>$O/bench_ab.err | tail -1 > $O/bench_f${fl}_$i.json
done; done
timeout 600 python bench.py --dtype fp16 --no-cpu-baseline 2>$O/bench_fp16.err | tail -1 > $O/bench_fp16.json
python - <<'PY'
import glob, json
for f in ["gpurun_out/r04u/bench_T16.json", "gpurun_out/r04u/bench_fp16.json"] + sorted(glob.glob("gpurun_out/r04u/bench_f*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], r["dtype"], r["value"], "ms", r["ms_per_step"], "enc", r["encode_ms"], "pre", r["prefill_ms"], "dec", r["decode_ms_per_token"],
              "fwd", r.get("forward_mfma_frac"), "hbm", r.get("decode_hbm_frac"), "roof", r.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/stc_bench.txt

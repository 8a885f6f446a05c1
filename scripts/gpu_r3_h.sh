#!/bin/bash
# 192-row tile variant (gemm variant 12): GPU bit-identity tests, per-shape kernel_bench at T=16, two bench lines
O=gpurun_out/${1:-r03h}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -3 | tee $O/pytest_gemm.log
timeout 300 python scripts/kernel_bench.py --frames 16 2>&1 | grep -v amdgpu.ids | tee $O/kernel_bench_T16.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$i.json; python -c "import sys,json; j=json.loads(open('$O/bench_$i.json').read()); print(j['encode_ms'], j['prefill_ms'], j['decode_ms_per_token'], j['ms_per_step'], j['forward_mfma_frac'], j['roofline']['frac'])"; done

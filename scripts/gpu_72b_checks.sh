#!/bin/bash
# 72B extras: the 8192-channel connector vs the oracle, and the multi-rank control flow of `bench.py --model 72b --tp` (2 ranks
# over gloo sharing the one GPU: each holds half of the decoder; timing meaningless)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_v21.py -x -q -m gpu -k "72b" > gpurun_out/pytest_72b.log 2>&1; echo "pytest exit $?"
export VL2_DIST_BACKEND=gloo
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --model 72b --gpus 2 --tp --steps 1 --warmup 0 --new-tokens 4 > gpurun_out/bench_gloo_72b_tp2.json 2> gpurun_out/bench_gloo_72b_tp2.err
echo "72b tp2 exit $?"

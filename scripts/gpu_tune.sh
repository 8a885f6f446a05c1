#!/bin/bash
mkdir -p gpurun_out
for t in "2=1" "2=2" "2=4"; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tune $t 2>/dev/null | grep -o '"decode_ms_per_token": [0-9.]*' | sed "s/^/tune $t: /"
done
# also validate the torch.distributed (RCCL) launch path with a single rank
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400

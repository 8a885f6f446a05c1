#!/bin/bash
# Debug: N ranks sharing the ONE GPU of this box over gloo (host-staged collectives) -- exercises the multi-rank control
# flow of bench.py / dist.py / the tensor-parallel decoder end to end (deadlocks, rank-asymmetric code); timing numbers are
# meaningless (shared GPU).
set -u
mkdir -p gpurun_out
export VL2_DIST_BACKEND=gloo
for N in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) \
      bench.py --gpus $N --steps 2 --warmup 1 --new-tokens 8 > gpurun_out/bench_gloo_$N.json 2> gpurun_out/bench_gloo_$N.err
  echo "N=$N exit $?"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --tp --steps 2 --warmup 1 --new-tokens 8 > gpurun_out/bench_gloo_tp2.json 2> gpurun_out/bench_gloo_tp2.err
echo "tp2 exit $?"

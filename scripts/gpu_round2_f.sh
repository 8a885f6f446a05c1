#!/bin/bash
# Round 2, GPU call F: (1) bench.py --gpus 2 / 4 with several ranks SHARING the one GPU over gloo (debug backend: collectives
# staged through the host) -- exercises the multi-rank control flow of the head of the round end to end: rank-local encoder
# hipGraphs, halo exchange, all-gather, the bit-equality check against the unsharded encoder, TP decode; timings are meaningless
# (shared GPU).  (2) the direct Conv3d micro-benchmark.  (3) a default bench line.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r02f}
mkdir -p $O
timeout 120 scripts/ubench/conv3d_direct > $O/conv3d_direct.json 2> $O/conv3d_direct.err
export VL2_DIST_BACKEND=gloo
for N in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) \
      bench.py --gpus $N --steps 2 --warmup 1 --new-tokens 8 > $O/bench_gloo_$N.json 2> $O/bench_gloo_$N.err
  echo "N=$N exit $?" >> $O/status.txt
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --tp --steps 2 --warmup 1 --new-tokens 8 > $O/bench_gloo_tp2.json 2> $O/bench_gloo_tp2.err
echo "tp2 exit $?" >> $O/status.txt
unset VL2_DIST_BACKEND
timeout 600 python bench.py > $O/bench_T16.json 2> $O/bench_T16.err
echo done

#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "sharded" > gpurun_out/pytest_sharded.log 2>&1; echo "pytest exit $?"
timeout 900 python scripts/shard_model.py > gpurun_out/shard_model.jsonl 2> gpurun_out/shard_model.err; echo "shard exit $?"

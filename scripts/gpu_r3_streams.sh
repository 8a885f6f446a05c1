#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r03c}
mkdir -p $O
for S in 1 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --new-tokens 4 --vit-streams $S 2> $O/streams_$S.err | python -c "import sys, json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $S, 'encode_ms', j['encode_ms'], 'prefill_ms', j['prefill_ms'])"
done > $O/streams.txt 2>&1
cat $O/streams.txt

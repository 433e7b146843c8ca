#!/bin/bash
for a in "--inflight 1" "--inflight 2" "--inflight 3" "--batch 2" "--batch 2 --inflight 2"; do
  echo -n "$a: "; DI_GRAPH_NODES=0 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['batch_per_gpu'])"
done

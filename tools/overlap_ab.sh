#!/bin/bash
# A/B of the fork/join sites (DI_OVERLAP bit mask), one bench process each
for m in ${MASKS:-0 1 3 5 9 15}; do
  echo -n "DI_OVERLAP=$m "; DI_OVERLAP=$m DI_GRAPH_NODES=0 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done

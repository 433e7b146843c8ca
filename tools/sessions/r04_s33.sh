#!/bin/bash
# round 4, call 33: hardware queue count of the HIP runtime (GPU_MAX_HW_QUEUES, default 4) against the 4-6 concurrent branches
# of two / three captured forwards in flight
OUT=gpurun_out/r04zq; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
for q in 4 8 2; do for n in 2 3; do
( GPU_MAX_HW_QUEUES=$q $B --inflight $n > $OUT/q${q}_n$n.json ) 2> $OUT/q${q}_n$n.err
python - $OUT/q${q}_n$n.json q${q}_n$n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d.get('single_sample',{}).get('ms_per_step'))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done; done

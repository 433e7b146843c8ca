#!/bin/bash
# round 4, call 23: mixed-precision step with fp16 model weights + float32 masters (A/B against per-step autocast casts)
OUT=gpurun_out/r04zb; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --mode train --steps 10 --warmup 3"
( $B --amp > $OUT/train_graph_amp.json ) 2> $OUT/train_graph_amp.err
( DI_TRAIN_HALF_WEIGHTS=0 $B --amp > $OUT/train_graph_amp_casts.json ) 2> $OUT/train_graph_amp_casts.err
for f in train_graph_amp train_graph_amp_casts; do python - $OUT/$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['first_loss'], d['last_loss'], d['dtype'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
tail -n 4 $OUT/train_graph_amp.err
( time timeout 900 python -m pytest tests/test_training_gpu.py -q -x -k "graphed or autocast or full_training" 2>&1 ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log

#!/bin/bash
# round 4, call 22: training step after the GT-only targets moved under the forward replay and the IoU kernel
OUT=gpurun_out/r04za; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --mode train --steps 10 --warmup 3"
( DI_TRAIN_GRAPH=1 DI_TRAIN_AMP=1 $B > $OUT/train_graph_amp.json ) 2> $OUT/train_graph_amp.err
( DI_TRAIN_GRAPH=1 $B > $OUT/train_graph.json ) 2> $OUT/train_graph.err
( $B > $OUT/train.json ) 2> $OUT/train.err
for f in train_graph_amp train_graph train; do python - $OUT/$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['first_loss'], d['last_loss'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
timeout 200 python tools/loss_cprofile.py 12 2>&1 | head -24
( time timeout 900 python -m pytest tests/test_training_gpu.py tests/test_targets_loss.py -q -x 2>&1 ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log

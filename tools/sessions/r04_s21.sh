#!/bin/bash
# round 4, call 21: the training step with the fused mixed-precision window attention (bench lines first: fresh find-db)
OUT=gpurun_out/r04z; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --mode train --steps 10 --warmup 3"
( DI_TRAIN_GRAPH=1 DI_TRAIN_AMP=1 $B > $OUT/train_graph_amp.json ) 2> $OUT/train_graph_amp.err
( DI_TRAIN_GRAPH=1 DI_TRAIN_AMP=1 DI_TRAIN_FUSED_LA=0 $B > $OUT/train_graph_amp_unfused.json ) 2> $OUT/train_graph_amp_unfused.err
( DI_TRAIN_AMP=1 $B > $OUT/train_amp.json ) 2> $OUT/train_amp.err
for f in train_graph_amp train_graph_amp_unfused train_amp; do python - $OUT/$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
( time timeout 900 python -m pytest tests/test_training_gpu.py tests/test_local_attn_train_gpu.py -q -x 2>&1 ) > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
for f in $OUT/*.err; do echo "== $f"; tail -n 3 $f; done

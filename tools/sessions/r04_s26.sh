#!/bin/bash
# round 4, call 26: fused training BatchNorm + ReLU
OUT=gpurun_out/r04zg; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --mode train --steps 10 --warmup 3"
( $B --amp > $OUT/train_amp.json ) 2> $OUT/train_amp.err
( DI_TRAIN_FUSED_BN=0 $B --amp > $OUT/train_amp_miopen_bn.json ) 2> $OUT/train_amp_miopen_bn.err
( $B > $OUT/train_f32.json ) 2> $OUT/train_f32.err
for f in train_amp train_amp_miopen_bn train_f32; do python - $OUT/$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['first_loss'], d['last_loss'], d['dtype'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
tail -n 3 $OUT/train_amp.err
( time timeout 900 python -m pytest tests/test_training_gpu.py -q -x 2>&1 ) > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log

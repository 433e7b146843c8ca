#!/bin/bash
# round 4, call 30: slabbed GEMM gradients in the decoder's training path (A/B/A/B on one box)
OUT=gpurun_out/r04zm; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --mode train --steps 10 --warmup 3 --amp"
git_stash() { :; }
for i in 1 2; do
( $B > $OUT/new_$i.json ) 2> $OUT/new_$i.err
( DI_TRAIN_PLAIN_GEMMS=1 $B > $OUT/old_$i.json ) 2> $OUT/old_$i.err
done
for f in new_1 old_1 new_2 old_2; do python - $OUT/$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['last_loss'])
PY
done
( timeout 600 python -m pytest tests/test_training_gpu.py tests/test_decoder_gpu.py -q -x 2>&1 ) | tail -2

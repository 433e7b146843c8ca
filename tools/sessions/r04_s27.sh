#!/bin/bash
# round 4, call 27: A/B/A/B of the fused training BatchNorm on ONE box (boxes differ by up to 10 %)
OUT=gpurun_out/r04zh; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --mode train --steps 10 --warmup 3 --amp"
for i in 1 2; do
( $B > $OUT/fused_$i.json ) 2> $OUT/fused_$i.err
( DI_TRAIN_FUSED_BN=0 $B > $OUT/miopen_$i.json ) 2> $OUT/miopen_$i.err
done
for f in fused_1 miopen_1 fused_2 miopen_2; do python - $OUT/$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])
PY
done

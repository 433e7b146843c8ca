#!/bin/bash
# round 4, call 31: the two image-side window attentions of a layer on two streams (DI_OVERLAP bit 6), A/B/A/B on one box
OUT=gpurun_out/r04zo; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0"
for i in 1 2; do
( DI_OVERLAP=93 $B > $OUT/two_streams_$i.json ) 2> $OUT/two_streams_$i.err
( DI_OVERLAP=29 $B > $OUT/one_stream_$i.json ) 2> $OUT/one_stream_$i.err
done
for f in two_streams_1 one_stream_1 two_streams_2 one_stream_2; do python - $OUT/$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['single_sample']['ms_per_step'], d['config']['graph_nodes'])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
tail -n 2 $OUT/two_streams_1.err

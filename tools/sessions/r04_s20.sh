#!/bin/bash
# round 4, call 20: phase-locked lanes against free-running lanes (headline and ++), and the new graph test
OUT=gpurun_out/r04x; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
( $B > $OUT/locked2.json ) 2> $OUT/locked2.err
( $B --free-running > $OUT/free2.json ) 2> $OUT/free2.err
( $B --inflight 3 > $OUT/locked3.json ) 2> $OUT/locked3.err
( $B --inflight 3 --free-running > $OUT/free3.json ) 2> $OUT/free3.err
( $B --inflight 4 > $OUT/locked4.json ) 2> $OUT/locked4.err
( $B --model pp > $OUT/pp_locked2.json ) 2> $OUT/pp_locked2.err
( $B --model pp --inflight 3 > $OUT/pp_locked3.json ) 2> $OUT/pp_locked3.err
for f in locked2 free2 locked3 free3 locked4 pp_locked2 pp_locked3; do python - $OUT/$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d.get('single_sample'))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
( time timeout 900 python -m pytest tests/test_graph_gpu.py -q -x 2>&1 ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
tail -3 $OUT/*.err | tail -40

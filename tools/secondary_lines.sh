#!/bin/bash
# gpurun: the secondary bench lines on the current code -> gpurun_out/TAG/*.json (copy into profiles/ as rNN_bench_*.json)
TAG=${1:-sec}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; python - "$OUT/$name.json" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['value'], d['unit'], d['ms_per_step'], d.get('single_sample', {}).get('ms_per_step'), d['config'].get('graph_nodes'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
P
}
# (the training lines first: MIOpen's user find-db is still empty then)
run bench_mode_train_amp --steps 10 --warmup 3 --mode train --amp
run bench_mode_train --steps 10 --warmup 3 --mode train
run bench_mode_train_eager --steps 10 --warmup 3 --mode train --train-eager
run bench_inflight3 --steps 20 --warmup 5 --inflight 3
run bench_from_raw --steps 20 --warmup 5 --from-raw
run bench_from_points --steps 20 --warmup 5 --from-points
run bench_model_pp --steps 20 --warmup 5 --model pp
run bench_shapeA --steps 20 --warmup 5 --shape A
echo done

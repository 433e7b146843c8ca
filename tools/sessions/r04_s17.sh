#!/bin/bash
OUT=gpurun_out/r04q; mkdir -p $OUT; export TMPDIR=/tmp
echo "== graphed train test"; timeout 600 python -m pytest tests/test_training_gpu.py -q -x -k "graphed or dropout or full_training" > $OUT/t.log 2>&1; tail -15 $OUT/t.log
echo "== train graphed bench"; DI_TRAIN_GRAPH=1 timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $OUT/train_graph.json 2> $OUT/train_graph.err; python -c "
import json; r=json.load(open('$OUT/train_graph.json')); print(r['value'], r['ms_per_step'], r['first_loss'], r['last_loss'])"

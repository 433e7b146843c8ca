#!/bin/bash
OUT=gpurun_out/r04p; mkdir -p $OUT; export TMPDIR=/tmp
echo "== train eager"; timeout 300 python bench.py --mode train --steps 10 --warmup 3 > $OUT/train_eager.json 2> $OUT/train_eager.err; tail -c 400 $OUT/train_eager.json; tail -2 $OUT/train_eager.err
echo "== train graphed"; DI_TRAIN_GRAPH=1 timeout 300 python bench.py --mode train --steps 10 --warmup 3 > $OUT/train_graph.json 2> $OUT/train_graph.err; tail -c 400 $OUT/train_graph.json; grep -v "Warn\|amdgpu" $OUT/train_graph.err | tail -25

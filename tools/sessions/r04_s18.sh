#!/bin/bash
OUT=gpurun_out/r04r; mkdir -p $OUT; export TMPDIR=/tmp
echo "== graph tests"; timeout 600 python -m pytest tests/test_graph_gpu.py -q -x > $OUT/t.log 2>&1; tail -3 $OUT/t.log
echo "== bench"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; r=json.load(open('$OUT/bench.json')); print(r['value'], r['ms_per_step'], r['single_sample'])"
echo "== bench pp"; timeout 300 python bench.py --model pp --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_pp.json 2> $OUT/bench_pp.err; python -c "
import json; r=json.load(open('$OUT/bench_pp.json')); print(r['value'], r['ms_per_step'])"

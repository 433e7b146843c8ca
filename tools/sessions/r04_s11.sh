#!/bin/bash
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp
for v in 0 40 41; do
echo "== bench AUTO=$v"; DI_LA_AUTO=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench_auto$v.json 2> $OUT/bench_auto$v.err; python - $OUT/bench_auto$v.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(r['value'], r['ms_per_step'], r['single_sample'], r['roofline']['avg_launch_us'], r['roofline']['frac'], r['roofline']['in_step_avg_us'])
PY
done

#!/bin/bash
OUT=gpurun_out/r04s; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pp tests"; timeout 900 python -m pytest tests/test_plusplus_gpu.py tests/test_shapePP_parity_gpu.py -q -x > $OUT/pp_tests.log 2>&1; tail -4 $OUT/pp_tests.log
cp gpurun_out/parity_shapePP.json $OUT/ 2>/dev/null
for ov in 29 28; do
echo "== bench pp DI_OVERLAP=$ov"; DI_OVERLAP=$ov timeout 600 python bench.py --model pp --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_pp_ov$ov.json 2> $OUT/bench_pp_ov$ov.err; python -c "
import json; r=json.load(open('$OUT/bench_pp_ov$ov.json')); print(r['value'], r['ms_per_step'], r['config']['graph_nodes'])"; tail -2 $OUT/bench_pp_ov$ov.err | grep -v amdgpu
done

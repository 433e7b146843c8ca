#!/bin/bash
OUT=gpurun_out/r04n; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pp fused head test"; timeout 600 python -m pytest tests/test_plusplus_gpu.py -q -x -k "head or quirk" > $OUT/pp_head.log 2>&1; tail -25 $OUT/pp_head.log
echo "== token tests"; timeout 600 python -m pytest tests/test_token_gpu.py tests/test_decoder_gpu.py -q -x > $OUT/tok.log 2>&1; tail -3 $OUT/tok.log
echo "== shapePP parity"; timeout 1200 python -m pytest tests/test_shapePP_parity_gpu.py -q > $OUT/pp_parity.log 2>&1; tail -8 $OUT/pp_parity.log
cp gpurun_out/parity_shapePP.json $OUT/ 2>/dev/null
echo "== bench pp"; timeout 900 python bench.py --model pp --steps 20 --warmup 5 > $OUT/bench_model_pp.json 2> $OUT/bench_model_pp.err; tail -c 600 $OUT/bench_model_pp.json | head -c 300; echo; python -c "
import json; r=json.load(open('$OUT/bench_model_pp.json')); print(r['value'], r['ms_per_step'], r['config']['graph_nodes']); p=r['parity']; print({k:(v['max'],v['p999']) for k,v in p.items() if k.startswith('dec.')})"; tail -3 $OUT/bench_model_pp.err

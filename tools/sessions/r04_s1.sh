#!/bin/bash
# round 4, session 1: ring window-attention generation (timing + bit identity), ++ full-shape parity, conditioned head, ++ bench lines
OUT=gpurun_out/r04a; mkdir -p $OUT; export TMPDIR=/tmp
echo "== la_bench2 img"; LA_SHAPE=img timeout 300 python tools/la_bench2.py 4 24 25 26 27 > $OUT/la_img.txt 2>&1; tail -8 $OUT/la_img.txt
echo "== la_bench2 bev"; LA_SHAPE=bev timeout 300 python tools/la_bench2.py 4 24 25 26 27 > $OUT/la_bev.txt 2>&1; tail -8 $OUT/la_bev.txt
echo "== ring tests"; timeout 600 python -m pytest tests/test_encoder_gpu.py -q -x -k "ring or kernel_variants" > $OUT/ring_tests.log 2>&1; tail -5 $OUT/ring_tests.log
echo "== shapePP parity"; timeout 1200 python -m pytest tests/test_shapePP_parity_gpu.py -q > $OUT/pp_parity.log 2>&1; tail -30 $OUT/pp_parity.log
cp gpurun_out/parity_shapePP.json $OUT/ 2>/dev/null
echo "== conditioned head"; timeout 900 python -m pytest tests/test_shapeR_parity_gpu.py -q -k "conditioned or fp16_eager" > $OUT/cond.log 2>&1; tail -15 $OUT/cond.log
cp gpurun_out/parity_shapeR.json $OUT/ 2>/dev/null
echo "== bench pp"; timeout 900 python bench.py --model pp --steps 20 --warmup 5 > $OUT/bench_model_pp.json 2> $OUT/bench_model_pp.err; tail -c 3000 $OUT/bench_model_pp.json; tail -5 $OUT/bench_model_pp.err
echo "== bench pp from images"; timeout 600 python bench.py --model pp --from-images --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_model_pp_from_images.json 2> $OUT/bench_model_pp_from_images.err; tail -c 1200 $OUT/bench_model_pp_from_images.json; tail -5 $OUT/bench_model_pp_from_images.err
echo done

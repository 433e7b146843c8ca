#!/bin/bash
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
echo "== dma_rate"; timeout 120 tools/micro/dma_rate > $OUT/dma_rate.txt 2>&1; cat $OUT/dma_rate.txt
echo "== la_bench2 img"; LA_SHAPE=img timeout 300 python tools/la_bench2.py 4 26 28 29 30 > $OUT/la_img.txt 2>&1; tail -6 $OUT/la_img.txt
echo "== la_bench2 bev"; LA_SHAPE=bev timeout 300 python tools/la_bench2.py 4 26 28 29 30 > $OUT/la_bev.txt 2>&1; tail -6 $OUT/la_bev.txt
echo "== ring tests"; timeout 600 python -m pytest tests/test_encoder_gpu.py -q -x -k "ring" > $OUT/ring_tests.log 2>&1; tail -3 $OUT/ring_tests.log
echo "== shapePP parity"; timeout 1200 python -m pytest tests/test_shapePP_parity_gpu.py -q > $OUT/pp_parity.log 2>&1; tail -8 $OUT/pp_parity.log
cp gpurun_out/parity_shapePP.json $OUT/ 2>/dev/null
echo "== grad parity shape R"; timeout 900 python -m pytest tests/test_training_gpu.py -q -k "shape_R" > $OUT/grad.log 2>&1; tail -15 $OUT/grad.log
cp gpurun_out/grad_parity_shapeR.json $OUT/ 2>/dev/null
echo "== bench B=2 x 2 in flight"; timeout 600 python bench.py --batch 2 --inflight 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_b2_i2.json 2> $OUT/bench_b2_i2.err; tail -c 1500 $OUT/bench_b2_i2.json; tail -3 $OUT/bench_b2_i2.err
echo "== bench B=2 x 1"; timeout 600 python bench.py --batch 2 --inflight 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_b2_i1.json 2> $OUT/bench_b2_i1.err; tail -c 600 $OUT/bench_b2_i1.json; tail -3 $OUT/bench_b2_i1.err
echo done

#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
echo "== la_bench2 img"; LA_SHAPE=img timeout 300 python tools/la_bench2.py 4 24 25 26 27 28 29 30 > $OUT/la_img.txt 2>&1; tail -9 $OUT/la_img.txt
echo "== la_bench2 bev"; LA_SHAPE=bev timeout 300 python tools/la_bench2.py 4 24 25 26 27 28 29 30 > $OUT/la_bev.txt 2>&1; tail -9 $OUT/la_bev.txt
echo "== ring tests"; timeout 600 python -m pytest tests/test_encoder_gpu.py -q -x -k "ring" > $OUT/ring_tests.log 2>&1; tail -3 $OUT/ring_tests.log

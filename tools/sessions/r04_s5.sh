#!/bin/bash
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ring tests"; timeout 600 python -m pytest tests/test_encoder_gpu.py -q -x -k "ring" > $OUT/ring_tests.log 2>&1; tail -3 $OUT/ring_tests.log
for pfd in 5 7 9 12 16; do
echo "== la_bench2 img pfd $pfd"; DI_RING_PFD=$pfd LA_SHAPE=img timeout 300 python tools/la_bench2.py 4 26 31 32 33 > $OUT/la_img_pfd$pfd.txt 2>&1; tail -5 $OUT/la_img_pfd$pfd.txt
done
echo "== la_bench2 bev"; LA_SHAPE=bev timeout 300 python tools/la_bench2.py 4 26 31 32 33 > $OUT/la_bev.txt 2>&1; tail -5 $OUT/la_bev.txt

#!/bin/bash
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ring tests"; timeout 600 python -m pytest tests/test_encoder_gpu.py -q -x -k "ring" > $OUT/ring_tests.log 2>&1; tail -3 $OUT/ring_tests.log
echo "== la_bench2 img"; LA_SHAPE=img timeout 300 python tools/la_bench2.py 4 26 34 35 36 37 > $OUT/la_img.txt 2>&1; tail -6 $OUT/la_img.txt
echo "== la_bench2 bev"; LA_SHAPE=bev timeout 300 python tools/la_bench2.py 4 26 34 35 36 37 > $OUT/la_bev.txt 2>&1; tail -6 $OUT/la_bev.txt
for dbg in 1 8 9; do
echo "== dbg $dbg"; DI_RING_DBG=$dbg LA_SHAPE=img timeout 300 python tools/la_bench2.py 34 > $OUT/la_dbg$dbg.txt 2>&1; tail -1 $OUT/la_dbg$dbg.txt
done

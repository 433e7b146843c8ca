#!/bin/bash
OUT=gpurun_out/r04i; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ring tests"; timeout 600 python -m pytest tests/test_encoder_gpu.py -q -x -k "ring" > $OUT/ring_tests.log 2>&1; tail -3 $OUT/ring_tests.log
for pfd in 1 2; do
echo "== la_bench2 img pfd(tiles) $pfd"; DI_RING_PFD=$pfd LA_SHAPE=img timeout 300 python tools/la_bench2.py 4 26 38 39 40 41 42 > $OUT/la_img_pfd$pfd.txt 2>&1; tail -7 $OUT/la_img_pfd$pfd.txt
done
echo "== la_bench2 bev"; DI_RING_PFD=1 LA_SHAPE=bev timeout 300 python tools/la_bench2.py 4 26 38 40 41 42 > $OUT/la_bev.txt 2>&1; tail -6 $OUT/la_bev.txt
DI_RING_DBG=16 timeout 120 python tools/ring_timeline.py 40 > $OUT/timeline_40.txt 2>&1; head -c 1500 $OUT/timeline_40.txt

#!/bin/bash
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
for dbg in 0 1 2 3 4 8 12 7 15; do
echo "== dbg $dbg"; DI_RING_DBG=$dbg LA_SHAPE=img timeout 300 python tools/la_bench2.py 26 25 > $OUT/la_dbg$dbg.txt 2>&1; tail -2 $OUT/la_dbg$dbg.txt
done

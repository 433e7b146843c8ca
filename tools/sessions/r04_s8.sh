#!/bin/bash
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
DI_RING_DBG=16 timeout 120 python tools/ring_timeline.py 26 > $OUT/timeline_26.txt 2>&1; head -c 6000 $OUT/timeline_26.txt
DI_RING_DBG=25 timeout 120 python tools/ring_timeline.py 26 > $OUT/timeline_26_nocompute_nostore.txt 2>&1

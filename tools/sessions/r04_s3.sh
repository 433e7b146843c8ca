#!/bin/bash
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/debug/grad_path_forward.py > $OUT/grad_dbg.txt 2>&1; grep -v Warn $OUT/grad_dbg.txt | tail -20

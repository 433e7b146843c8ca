#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py with the given flags: tools/prof_flags.sh TAG NAME flags...   -> gpurun_out/TAG/kernel_stats_NAME.csv
TAG=$1; NAME=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$NAME -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" ) > $OUT/rocprof_$NAME.log 2>&1
find $OUT/prof_$NAME -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$NAME.csv; rm -rf $OUT/prof_$NAME
tail -1 $OUT/rocprof_$NAME.log | cut -c1-300

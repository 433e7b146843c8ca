#!/bin/bash
# gpurun: serial kernel trace of the benched forward (one sample at a time, one stream) -> gpurun_out/TAG/{kernel_stats_serial.csv,seq.txt}
TAG=${1:-t}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && DI_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline ${BENCH_ARGS} ) > $OUT/rocprof.log 2>&1
tail -c 400 $OUT/rocprof.log
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial.csv
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_seq.py $T ${ANCHOR:-heatmap_nms} > $OUT/seq.txt 2>&1
rm -rf $OUT/prof
tail -80 $OUT/seq.txt

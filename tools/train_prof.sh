#!/bin/bash
# gpurun: kernel trace of the training step; steady-state per-kernel table (last 3 steps) -> gpurun_out/TAG/train_tail.txt
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-tr}; mkdir -p $OUT
cd /tmp; rm -rf /tmp/trp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $OUT/train.json 2> $OUT/train.err
MS=$(python -c "import json;print(json.loads(open('$OUT/train.json').read().strip().splitlines()[-1])['ms_per_step'])")
f=$(find /tmp/trp -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/trace_tail_stats.py $f $MS 3 60 > $OUT/train_tail.txt
head -c 200 $OUT/train.json | tail -c 120; echo; cat $OUT/train_tail.txt

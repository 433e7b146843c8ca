#!/bin/bash
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-tr}; mkdir -p $OUT
cd /tmp; rm -rf /tmp/trp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline > $OUT/train.json 2> $OUT/train.err
tail -c 300 $OUT/train.json
f=$(find /tmp/trp -name '*kernel_stats.csv' | head -1); cp $f $OUT/train_kernel_stats.csv

#!/bin/bash
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pp}; mkdir -p $OUT
python bench.py --model pp --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_pp.json 2> $OUT/bench_pp.err; tail -c 400 $OUT/bench_pp.json
cd /tmp; rm -rf /tmp/ppp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ppp -o t -- python $GRAFT_REPO_ROOT/bench.py --model pp --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rocprof_pp.log 2>&1
f=$(find /tmp/ppp -name '*kernel_stats.csv' | head -1); cp $f $OUT/pp_kernel_stats.csv

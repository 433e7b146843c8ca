#!/bin/bash
# gpurun: rocprofv3 kernel trace of tools/i2p_bench.py (both attention passes + the key build); prints the i2p kernels' rows.
# Usage: tools/i2p_prof.sh TAG   (environment such as DI_I2PD_NB / DI_I2PD_BLOCKS is passed through)
TAG=${1:-i2p}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o i2p -- python $GRAFT_REPO_ROOT/tools/i2p_bench.py ) > $OUT/rocprof.log 2>&1
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv
rm -rf $OUT/prof
grep -i "i2p\|Name" $OUT/kernel_stats.csv | cut -c1-200
tail -4 $OUT/rocprof.log

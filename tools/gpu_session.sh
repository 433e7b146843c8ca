#!/bin/bash
# One gpurun session: GPU parity tests, bench line, rocprofv3 kernel stats, PMC passes.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh <tag>
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json
timeout 300 python tools/kernel_bench.py > $OUT/kernel_bench.txt 2>&1; cat $OUT/kernel_bench.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o stats -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pp_prof -o stats -- python $REPO/tools/pp_bench.py --steps 10 > $OUT/pp_bench.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pp_pmc_fetch -o pmc -- python $REPO/tools/pp_bench.py --steps 2 > $OUT/pp_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pp_pmc_write -o pmc -- python $REPO/tools/pp_bench.py --steps 2 > $OUT/pp_pmc_write.log 2>&1
cd $REPO
grep -E "forward|fwd" $OUT/pp_bench.txt | head -12
timeout 200 python tools/train_bench.py --dtype f32 --steps 5 > $OUT/train_bench.txt 2>&1; tail -1 $OUT/train_bench.txt
timeout 200 python tools/pp_train_bench.py --dtype f32 --steps 3 > $OUT/pp_train_bench.txt 2>&1; tail -1 $OUT/pp_train_bench.txt
find $OUT -name '*.csv' | head -30
# keep the merge-back small: drop full kernel traces above 20 MB
find $OUT -name '*.csv' -size +20M -delete

#!/bin/bash
# One gpurun call.  Usage: [PRE="tests..."] [SKIP_FULL=1] [QUICK=1] [EXTRA="cmd"] tools/gpu_session.sh TAG
# QUICK=1 skips the second bench run and the two-samples-in-flight kernel trace (saves about 40 s of box time).
# Order matters: the bench runs FIRST, in a fresh process on the fresh box (MIOpen's user find-db is empty then).
TAG=${1:-s}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench (driver flags)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 3500 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
if [ -z "$QUICK" ]; then echo "== bench (defaults, no cpu)"; timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json; fi
if [ -n "$EXTRA" ]; then echo "== extra: $EXTRA"; ( eval "$EXTRA" ) > $OUT/extra.log 2>&1; tail -40 $OUT/extra.log; fi
echo "== rocprof (serial: one sample at a time, one stream - per-kernel durations without time-sharing)"; ( cd /tmp && DI_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_serial -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline ) > $OUT/rocprof_serial.log 2>&1; find $OUT/prof_serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial.csv; rm -rf $OUT/prof_serial
if [ -z "$QUICK" ]; then echo "== rocprof"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; tail -2 $OUT/rocprof.log
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -type f ! -name '*stats.csv' -delete 2>/dev/null
fi
if [ -n "$PRE" ]; then echo "== pre: $PRE"; ( timeout 900 python -m pytest $PRE -x -q ) > $OUT/pre.log 2>&1; tail -25 $OUT/pre.log; fi
if [ -z "$SKIP_FULL" ]; then
echo "== pytest" ; ( time timeout 1500 python -m pytest tests -m gpu -x -q "$@" ) > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
fi
echo done

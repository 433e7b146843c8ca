#!/bin/bash
# ONE gpurun call = one invocation of this script:   tools/session.sh TAG step [step ...]
# (replaces the 29 one-off tools/sessions/r04_s*.sh of round 4).  Everything goes to gpurun_out/TAG/; every step has its own
# timeout so that a hung kernel cannot eat the box.  Steps:
#   bench                 bench.py with the driver's flags                                   -> bench_driver.json
#   bench:<name>:<flags>  another bench line (flags with '+' for spaces), no CPU baseline    -> bench_<name>.json
#   prof                  rocprofv3 --kernel-trace --stats of the serial forward             -> kernel_stats_serial.csv
#   prof2                 the same of the default (two in flight)                            -> kernel_stats.csv
#   tests:<pytest args>   pytest -m gpu subset ('+' for spaces; -x -q added)                 -> pytest_<n>.log
#   fulltests             the whole GPU suite + smoke()                                      -> pytest.log, smoke.log
#   py:<script+args>      python <script> ('+' for spaces)                                   -> py_<n>.log
#   sh:<command>          a shell command ('+' for spaces)                                   -> sh_<n>.log
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
n=0
line() { python - "$1" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(sys.argv[1], d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'single', d.get('single_sample', {}).get('ms_per_step'),
          'nodes', d['config'].get('graph_nodes'), 'frac', r.get('frac'), 'us', r.get('avg_launch_us'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
P
}
for step in "$@"; do
  n=$((n+1)); kind=${step%%:*}; arg=${step#*:}; arg=${arg//+/ }
  echo "== [$n] $step"
  case $kind in
    bench)
      if [ "$step" = bench ]; then
        timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; line $OUT/bench_driver.json; tail -2 $OUT/bench_driver.err
      else
        name=${arg%%:*}; flags=${arg#*:}
        timeout 600 python bench.py $flags --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; line $OUT/bench_$name.json; tail -2 $OUT/bench_$name.err
      fi ;;
    prof)
      ( cd /tmp && DI_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_serial -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline ${PROF_FLAGS} ) > $OUT/rocprof_serial.log 2>&1
      find $OUT/prof_serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial.csv; rm -rf $OUT/prof_serial; python tools/kernel_table.py $OUT/kernel_stats_serial.csv 30 ;;
    prof2)
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1
      find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; rm -rf $OUT/prof ;;
    tests)
      ( eval "time timeout 1500 python -m pytest $arg -x -q -m gpu" ) > $OUT/pytest_$n.log 2>&1; tail -25 $OUT/pytest_$n.log ;;
    fulltests)
      ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log ;;
    py)
      ( timeout 900 python $arg ) > $OUT/py_$n.log 2>&1; tail -40 $OUT/py_$n.log ;;
    sh)
      ( timeout 900 bash -c "$arg" ) > $OUT/sh_$n.log 2>&1; tail -40 $OUT/sh_$n.log ;;
    *) echo "unknown step $step" ;;
  esac
done
echo done

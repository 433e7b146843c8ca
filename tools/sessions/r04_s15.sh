#!/bin/bash
OUT=gpurun_out/r04o; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pp tests"; timeout 600 python -m pytest tests/test_plusplus_gpu.py -q -x > $OUT/pp_tests.log 2>&1; tail -4 $OUT/pp_tests.log
echo "== rocprof pp serial"; ( cd /tmp && DI_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_pp -o bench -- python $GRAFT_REPO_ROOT/bench.py --model pp --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 --settle-ms 0 ) > $OUT/rocprof_pp.log 2>&1; find $OUT/prof_pp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/pp_kernel_stats_serial.csv; find $OUT/prof_pp -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/pp_kernel_trace.csv; rm -rf $OUT/prof_pp; ls -la $OUT

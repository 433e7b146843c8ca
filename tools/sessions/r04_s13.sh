#!/bin/bash
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
echo "== grad parity shape R"; timeout 900 python -m pytest tests/test_training_gpu.py -q -k "shape_R" > $OUT/grad.log 2>&1; tail -5 $OUT/grad.log
cp gpurun_out/grad_parity_shapeR.json $OUT/ 2>/dev/null
echo "== rocprof pp serial"; ( cd /tmp && DI_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_pp -o bench -- python $GRAFT_REPO_ROOT/bench.py --model pp --steps 20 --warmup 5 --no-cpu-baseline ) > $OUT/rocprof_pp.log 2>&1; find $OUT/prof_pp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/pp_kernel_stats_serial.csv; rm -rf $OUT/prof_pp; tail -2 $OUT/rocprof_pp.log | cut -c1-600

#!/bin/bash
# gpurun: hardware counters of the kernels INSIDE the benched forward (eager launches, one stream, one sample at a time:
# every dispatch is its own kernel, so the counters are per launch).  Separate rocprofv3 --pmc passes with --kernel-trace
# only (gpurun refuses --pmc with other trace domains).  Usage: [MODEL=pp] tools/pmc_forward.sh TAG
TAG=${1:-pmcf}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--eager --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --roofline-steps 0"
[ -n "$MODEL" ] && ARGS="$ARGS --model $MODEL"
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  DI_OVERLAP=0 timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_forward_summary.py $f >> $OUT/rows.jsonl
  rm -rf $OUT/p$i
done
python $GRAFT_REPO_ROOT/tools/pmc_forward_summary.py --merge $OUT/rows.jsonl $TAG > $OUT/summary.json
cat $OUT/summary.json | head -60

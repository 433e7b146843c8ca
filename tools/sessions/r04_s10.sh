#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*\|SQ_[A-Z0-9_]*" $OUT/counters.txt | sort -u > $OUT/counter_names.txt; wc -l $OUT/counter_names.txt
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/la_eager.py 4 26 40 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'][:60]
    if 'local_attn' not in k: continue
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
  else tail -5 $OUT/p$i.log; fi
  rm -rf $OUT/p$i
done

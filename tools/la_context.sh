#!/bin/bash
# tools/la_context.sh TAG : the merged window-attention launch alone and behind three kinds of predecessors (tools/la_context.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-lactx}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in alone fill produce gemm alone; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -o t -- python $GRAFT_REPO_ROOT/tools/la_context.py $c > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['TotalDurationNs']) > 2e5:
        print('  %-70s calls %5s avg %8.2f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
P
  rm -rf $OUT/$c
done

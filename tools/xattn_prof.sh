#!/bin/bash
# kernel-trace timing of the decoder cross attention at the benched shape
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/xap; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xap -o t -- python $GRAFT_REPO_ROOT/tools/kernel_pmc.py xa > /tmp/xap.log 2>&1
f=$(find /tmp/xap -name '*kernel_stats.csv' | head -1)
python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mha_decode' in r['Name']:
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:7.2f} us  min {float(r['MinNs'])/1000:7.2f}")
P

#!/bin/bash
# kernel-trace timing of the window attention at the benched shapes (tools/kernel_pmc.py la)
export TMPDIR=/tmp; cd /tmp
for nt in 0 1 0 1; do
rm -rf /tmp/lap; DI_LA_NT=$nt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lap -o t -- python $GRAFT_REPO_ROOT/tools/kernel_pmc.py la > /tmp/lap.log 2>&1
f=$(find /tmp/lap -name '*kernel_trace.csv' | head -1)
echo "DI_LA_NT=$nt"; python - "$f" <<'P'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'local_attn' in r['Kernel_Name']:
        d[r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for g, v in d.items():
    print(f'  local_attn grid {g}:', ' '.join(f'{x:.1f}' for x in v))
P
done

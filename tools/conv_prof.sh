#!/bin/bash
# kernel-trace timing of the 3x3 convolution kernels at the benched shapes (tools/kernel_pmc.py conv)
export TMPDIR=/tmp; cd /tmp
for v in ${VARIANTS:-0 1}; do
rm -rf /tmp/cvp; DI_CONV_NO_DMA=0 DI_CONV_TH=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cvp -o t -- python $GRAFT_REPO_ROOT/tools/kernel_pmc.py conv > /tmp/cvp.log 2>&1
f=$(find /tmp/cvp -name '*kernel_stats.csv' | head -1)
echo "DI_CONV_TH=$v (0 = automatic)"; python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv3x3' in r['Name']:
        print(f"  {r['Name'][:48]:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:7.2f} min {float(r['MinNs'])/1000:7.2f} max {float(r['MaxNs'])/1000:7.2f}")
P
done

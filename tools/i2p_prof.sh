#!/bin/bash
# kernel-trace timing (not host-loop timing: a Python call costs more than these kernels) of the pillar attention
export TMPDIR=/tmp; cd /tmp
for b in ${BLOCKS:-2048}; do
  rm -rf /tmp/i2pp; DI_I2P_BLOCKS=$b timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/i2pp -o t -- python $GRAFT_REPO_ROOT/tools/i2p_bench.py > /tmp/i2pp.log 2>&1
  f=$(find /tmp/i2pp -name '*kernel_stats.csv' | head -1)
  echo "blocks $b"; python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'i2p' in r['Name']:
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:7.2f} us  min {float(r['MinNs'])/1000:7.2f}")
P
done

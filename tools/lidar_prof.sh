#!/bin/bash
# tools/lidar_prof.sh : rocprofv3 kernel trace of the frozen LiDAR branch alone (21 forwards at shape R), per-forward kernel table
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/lp
LIDAR_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o t -- python $GRAFT_REPO_ROOT/tools/lidar_backbone_bench.py > /tmp/lp.log 2>&1
f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].startswith("naive_conv")]   # MIOpen's solver timing runs its naive kernels once per shape: not part of a forward
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
n = 21.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e3
print("kernel us per forward %.1f, launches per forward %.1f" % (tot, sum(int(r["Calls"]) for r in rows) / n))
for r in rows[:28]:
    print("%-100s %6.1f calls avg %8.2f us  %8.1f us/fwd" % (r["Name"][:100], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / n))
P
grep "whole branch" /tmp/lp.log

#!/bin/bash
# gpurun: A/B of environment switches on ONE box, alternating, driver flags, no CPU baseline.
# Usage: tools/ab_bench.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...     (each argument = one configuration; '-' = defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1)); [ "$cfg" = "-" ] && cfg=""
    env $cfg timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/ab_${i}_$rep.json 2> $OUT/ab_${i}_$rep.err
    python - "$OUT/ab_${i}_$rep.json" "$cfg" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2] or 'defaults':40s} {d['value']:8.1f} samples/s  single {d['single_sample']['ms_per_step']} ms  copy {d['copy_handover']['value']}  nodes {d['config']['graph_nodes']}  la {d['roofline']['avg_launch_us']} us frac {d['roofline']['frac']}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
P
  done
done

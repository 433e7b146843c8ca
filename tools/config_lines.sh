#!/bin/bash
# gpurun: ONE BENCH-shaped JSON line per BASELINE.json configuration, each WITH `roofline` and (N = 1) `cpu_baseline`
#  -> gpurun_out/TAG/*.json (copy into profiles/ as rNN_*).  The training lines first: MIOpen's user
# find-db is still empty then.  Usage: tools/config_lines.sh TAG [quick]
TAG=${1:-cfg}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; ( time timeout 900 python bench.py "$@" ) > $OUT/$name.json 2> $OUT/$name.err; python - "$OUT/$name.json" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, c = d.get('roofline') or {}, d.get('cpu_baseline') or {}
    print(sys.argv[1], d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'single', (d.get('single_sample') or {}).get('ms_per_step'),
          'nodes', d['config'].get('graph_nodes'), '| roofline', r.get('avg_launch_us'), 'us frac', r.get('frac'), '| cpu', c.get('value'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
P
}
run cfg3_train_f32 --steps 10 --warmup 3 --mode train                      # configs[2] (N = 1) / [3] (N > 1), the reference's float32
run cfg3_train_amp --steps 10 --warmup 3 --mode train --amp
run cfg3_train_amp_b2 --steps 10 --warmup 3 --mode train --amp --batch 2   # the reference's samples_per_gpu = 2
run cfg5_pp_train_f32 --steps 5 --warmup 2 --mode train --model pp        # configs[4], training step (eager launches)
run cfg2_headline --gpus 1 --steps 20 --warmup 5                          # configs[1]: the driver's command
run cfg1_shapeA --steps 20 --warmup 5 --shape A                           # configs[0]'s maps through the full forward
run cfg5_pp_forward --steps 20 --warmup 5 --model pp                      # configs[4], forward
run sec_from_raw --steps 20 --warmup 5 --from-raw --no-cpu-baseline
run sec_from_points --steps 20 --warmup 5 --from-points --no-cpu-baseline
run sec_from_images --steps 20 --warmup 5 --from-images --no-cpu-baseline
run sec_from_sensors --steps 10 --warmup 3 --from-lidar --from-images --from-points --no-cpu-baseline   # raw points + camera images -> boxes' inputs
run sec_from_lidar --steps 10 --warmup 3 --from-lidar --no-cpu-baseline
echo done

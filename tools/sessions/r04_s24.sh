#!/bin/bash
# round 4, call 24: paired 16-byte output stores (v_permlane16_swap) in the ring, m2 and training window-attention kernels
OUT=gpurun_out/r04zd; mkdir -p $OUT; export TMPDIR=/tmp
( python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json ) 2> $OUT/bench_driver.err
python - $OUT/bench_driver.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('single_sample'), d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['in_step_avg_us'])
PY
( time timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_local_attn_train_gpu.py -q -x -k "local_att or training_attention" 2>&1 ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 200 python tools/la_bench2.py 4 24 > $OUT/la_img.txt 2>&1; tail -2 $OUT/la_img.txt
LA_SHAPE=bev timeout 200 python tools/la_bench2.py 4 24 > $OUT/la_bev.txt 2>&1; tail -2 $OUT/la_bev.txt
timeout 200 python tools/la_train_bench.py 2>&1 | tail -2

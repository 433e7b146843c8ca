mkdir -p gpurun_out/r05zm
run() { name=$1; shift; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --settle-ms 50 "$@" > gpurun_out/r05zm/$name.json 2> gpurun_out/r05zm/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('gpurun_out/r05zm/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('handover'), d['config'].get('inflight'))" 2>&1 | tail -1)"; }
run eager --eager
run from_raw --from-raw
run from_points --from-points
run from_images --from-images
run batch2 --batch 2
run shapeA --shape A
run tiny --shape TINY --proposals 24
run inflight1 --inflight 1
run inflight2_copy --inflight 2 --handover copy
run threads0 --launch-threads 0
run pp_inflight1 --model pp --inflight 1
run pp_eager --model pp --eager
run pp_images --model pp --from-images
run f32 --dtype f32 --inflight 2

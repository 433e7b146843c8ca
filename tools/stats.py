"""Per-forward kernel time table from a gpurun_out/<tag>/kernel_stats.csv + its bench json lines."""
import csv, json, sys
tag = sys.argv[1]
for f in ('bench_driver', 'bench_default'):
    try:
        r = json.loads(open(f'gpurun_out/{tag}/{f}.json').read().strip().splitlines()[-1])
        print(f, r['value'], r['ms_per_step'], r['config'].get('graph_nodes'))
    except Exception as e:
        print(f, 'n/a', e)
rows = list(csv.DictReader(open(f'gpurun_out/{tag}/kernel_stats.csv')))
la = [r for r in rows if 'local_attn_m' in r['Name']]
nfwd = sum(int(r['Calls']) for r in la) / 6
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('forwards', nfwd, 'kernel ms per fwd', round(tot / 1e6 / nfwd, 3), 'launches per fwd', round(sum(int(r['Calls']) for r in rows) / nfwd, 1))
keys = ('conv3x3', 'pointwise_multi', 'pointwise_chain', 'local_attn', 'tl_rowblock', 'pred_head', 'dynconv', 'tok_mha', 'tl_wide', 'tl_splitk', 'tl_finish', 'roi_select', 'query_init', 'i2p', 'copyBuffer', 'tk::', 'dc_', 'bevwarp', 'depth_', 'mha_decode', 'Cijk', 'query_geometry', 'roi_align', 'heatmap_nms', 'at::native')
groups = {}
for r in rows:
    g = next((k for k in keys if k in r['Name']), 'other:' + r['Name'][:40])
    a = groups.setdefault(g, [0, 0]); a[0] += int(r['TotalDurationNs']); a[1] += int(r['Calls'])
for g, (t, c) in sorted(groups.items(), key=lambda x: -x[1][0]):
    if t / 1e3 / nfwd >= 1:
        print(f'{t / 1e3 / nfwd:8.1f} us/fwd {c / nfwd:6.1f} launches  avg {t / 1e3 / max(c, 1):6.1f}  {g}')
if len(sys.argv) > 2:
    for r in rows:
        if any(k in r['Name'] for k in sys.argv[2].split(',')):
            print(f"{r['Name'][:80]:80s} per fwd {int(r['Calls']) / nfwd:4.1f} avg {float(r['AverageNs']) / 1e3:7.1f} min {int(r['MinNs']) / 1e3:6.1f} max {int(r['MaxNs']) / 1e3:6.1f}")

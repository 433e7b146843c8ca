"""Per-forward kernel table from a rocprofv3 kernel_stats csv of bench.py: python tools/kernel_table.py CSV [top]
(forwards = launches of the image-side window attention / 4 for the v1 model; `--pp`: / launches of the ++ image self-attention)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
ring = [int(r['Calls']) for r in rows if 'local_attn_ring' in r['Name']]
msda = [int(r['Calls']) for r in rows if 'ms_deform_attn' in r['Name'] and 'bwd' not in r['Name']]
conv = [int(r['Calls']) for r in rows if 'conv3x3_pc_kernel<20' in r['Name']]      # once per v1 forward (the image conv)
n_fwd = conv[0] if conv else (ring[0] / 4 if ring else (sum(msda) / 6 if msda else 1))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'forwards {n_fwd:.1f}   kernel us / forward {tot / n_fwd / 1e3:.1f}   launches / forward {sum(int(r["Calls"]) for r in rows) / n_fwd:.1f}')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:top]:
    print(f"{float(r['TotalDurationNs']) / n_fwd / 1e3:8.1f} us/fwd {int(r['Calls']) / n_fwd:6.1f} calls {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")

"""Neighbours (in start order, same step) of every launch of a kernel whose name contains PATTERN in a rocprofv3 kernel trace.
Usage: python tools/trace_context.py kernel_trace.csv PATTERN [before=3] [after=2] [max=6]"""
import csv, sys
f, pat = sys.argv[1], sys.argv[2]
nb, na, mx = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 3), (4, 2), (5, 6)))
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) * 2 // 3:]                      # the last third: steady state
hits = [i for i, r in enumerate(rows) if pat in r['Kernel_Name']]
print(len(hits), 'launches in the last third')
for h in hits[:mx]:
    for j in range(max(0, h - nb), min(len(rows), h + na + 1)):
        r = rows[j]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{'>>' if j == h else '  '} {d:8.1f} us  grid {r.get('Grid_Size_X', '?'):>8s} wg {r.get('Workgroup_Size_X', '?'):>4s}  {r['Kernel_Name'][:120]}")
    print()

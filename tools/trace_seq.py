"""Ordered kernel sequence of ONE forward from a rocprofv3 kernel trace (…_kernel_trace.csv): name, duration, gap to the
previous kernel's end.  Usage: python tools/trace_seq.py trace.csv [anchor-substring=query_init] [which=-2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else 'heatmap_nms'
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
a, b = idx[which], idx[which + 1]
prev_end = None
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {r['Kernel_Name'][:90]}")
    prev_end = max(e, prev_end or 0)
print('span', (prev_end - t0) / 1e3, 'us,', b - a, 'kernels')

# mean duration per position over the last forwards whose kernel sequence equals the printed one
names = [r['Kernel_Name'] for r in rows[a:b]]
acc, cnt = [0.0] * len(names), 0
for j in range(len(idx) - 1):
    seg = rows[idx[j]:idx[j + 1]]
    if [r['Kernel_Name'] for r in seg] != names:
        continue
    cnt += 1
    for i, r in enumerate(seg):
        acc[i] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f'--- mean duration per position over {cnt} forwards with this sequence (sum {sum(acc) / max(cnt, 1):.1f} us)')
for i, n in enumerate(names):
    print(f'{i:4d} {acc[i] / max(cnt, 1):7.1f}  {n[:100]}')

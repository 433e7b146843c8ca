"""Idle intervals of the GPU in the last step(s) of a rocprofv3 kernel trace: where a step is not bound by its kernels.
Usage: python tools/trace_gaps.py kernel_trace.csv [min_gap_us=100] [fraction of the trace looked at, from the end = 0.25]"""
import csv, sys
f = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * (1 - frac)):]
t0 = int(rows[0]['Start_Timestamp'])
busy_end, idle, gaps = int(rows[0]['End_Timestamp']), 0.0, []
for i in range(1, len(rows)):
    s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    if s > busy_end:
        g = (s - busy_end) / 1e3
        idle += g
        if g >= thr:
            gaps.append((g, (busy_end - t0) / 1e6, rows[i - 1]['Kernel_Name'][:70], rows[i]['Kernel_Name'][:70]))
    busy_end = max(busy_end, e)
span = (busy_end - t0) / 1e3
print(f'span {span / 1e3:.2f} ms, idle {idle / 1e3:.2f} ms ({100 * idle / span:.0f} %), {len(gaps)} gaps >= {thr} us = {sum(g[0] for g in gaps) / 1e3:.2f} ms')
for g, at, a, b in gaps:
    print(f'{g:8.0f} us at {at:7.2f} ms   after {a}\n{"":27s}before {b}')

"""How a several-lanes-in-flight step fills the GPU: from a rocprofv3 kernel trace, over the steady-state tail of the run - the share
of wall time with 0 / 1 / 2 / 3+ kernels running, and per kernel name the time it ran ALONE (nothing else on the chip) against
its total time.  Usage: python tools/trace_overlap.py kernel_trace.csv [fraction of the trace, from the end = 0.3] [rows = 25]"""
import collections, csv, sys
f = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows_n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * (1 - frac)):]
ev = []
for i, r in enumerate(rows):
    ev.append((int(r['Start_Timestamp']), 1, i))
    ev.append((int(r['End_Timestamp']), 0, i))
ev.sort()
active, last = set(), ev[0][0]
depth_t = collections.Counter()
alone, total = collections.Counter(), collections.Counter()
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        depth_t[min(len(active), 3)] += dt
        if len(active) == 1:
            alone[rows[next(iter(active))]['Kernel_Name']] += dt
        for j in active:
            total[rows[j]['Kernel_Name']] += dt
    last = t
    if kind:
        active.add(i)
    else:
        active.discard(i)
span = ev[-1][0] - ev[0][0]
print(f'{len(rows)} launches over {span / 1e6:.2f} ms; kernels running at once: ' +
      ', '.join(f'{k if k < 3 else "3+"}: {100 * v / span:.1f} %' for k, v in sorted(depth_t.items())))
print('alone ms   total ms   alone share   kernel')
for name, a in alone.most_common(rows_n):
    print(f'{a / 1e6:8.2f} {total[name] / 1e6:10.2f} {100 * a / total[name]:10.0f} %   {name[:100]}')

"""Per-kernel totals over the LAST `nsteps` steps of a traced run (steady state: the first steps carry MIOpen's solver
search).  Usage: python tools/trace_tail_stats.py kernel_trace.csv ms_per_step [nsteps=3] [top=45]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ms, n, top = float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3, int(sys.argv[4]) if len(sys.argv) > 4 else 45
end = max(int(r['End_Timestamp']) for r in rows)
t0 = end - int(n * ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r['Start_Timestamp']) >= t0:
        a = agg[r['Kernel_Name']]
        a[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a[1] += 1
tot = sum(a[0] for a in agg.values())
print(f'last {n} steps: kernel time {tot / n / 1e6:.2f} ms per step, {sum(a[1] for a in agg.values()) / n:.0f} launches per step')
for k, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:top]:
    print(f'{t / n / 1e3:9.1f} us/step {c / n:7.1f} launches  avg {t / c / 1e3:8.1f}  {k[:100]}')

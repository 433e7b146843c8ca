"""Whole-forward roofline summary = one committed JSON that `bench.py` quotes in `roofline.forward` / `roofline.kernels[*].pmc`:

    python tools/forward_roofline.py KERNEL_STATS_SERIAL_CSV PMC_FORWARD_SUMMARY_JSON TAG > profiles/TAG_forward_roofline.json

KERNEL_STATS_SERIAL_CSV: rocprofv3 --kernel-trace --stats of `DI_OVERLAP=0 bench.py --inflight 1` (tools/session.sh `prof`:
one kernel on the chip at a time, so durations are the kernels' own); PMC_FORWARD_SUMMARY_JSON: tools/pmc_forward.sh (separate
rocprofv3 --pmc passes over the EAGER forward: FETCH_SIZE | WRITE_SIZE | SQ busy counters).  Per kernel: launches per
forward, average duration, HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts
half of a wide streaming read on gfx950), MFMA / VALU / LDS busy.  Totals per forward: kernel time, HBM bytes, the average
HBM rate while a kernel runs and its fraction of the 8 TB/s peak."""
import csv
import json
import sys

HBM_PEAK = 8.0e12


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


def main():
    stats, pmc_path, tag = sys.argv[1:4]
    rows = list(csv.DictReader(open(stats)))
    ring = [int(r['Calls']) for r in rows if 'local_attn_ring' in r['Name']]
    msda = [int(r['Calls']) for r in rows if 'ms_deform_attn' in r['Name'] and 'bwd' not in r['Name']]
    conv = [int(r['Calls']) for r in rows if 'conv3x3_pc_kernel<20' in r['Name']]      # once per v1 forward (the image conv)
    n_fwd = conv[0] if conv else (ring[0] / 4 if ring else (sum(msda) / 6 if msda else 1))
    pmc = json.load(open(pmc_path))
    pk = pmc['kernels']
    by_name = {}
    for k, v in pk.items():
        by_name.setdefault(short(k.split(' grid=')[0]), []).append(v)
    kernels, tot_us, tot_bytes = [], 0.0, 0.0
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
        name = short(r['Name'])
        calls = int(r['Calls']) / n_fwd
        us = float(r['AverageNs']) / 1e3
        tot_us += float(r['TotalDurationNs']) / n_fwd / 1e3
        e = dict(kernel=name[:120], launches_per_forward=round(calls, 2), avg_us=round(us, 2))
        hits = by_name.get(name, [])
        if hits:
            # several grids of one kernel (image side / BEV side): launch-weighted mean of the per-launch figures
            n_l = b = 0.0
            busy = {}
            for h in hits:
                c = h['counters']
                n = (c.get('FETCH_SIZE') or next(iter(c.values())))['n']
                hb = h.get('hbm_bytes_per_launch_large')
                if hb is not None and h.get('hbm_bytes_per_launch_small') is not None:
                    cf = c['FETCH_SIZE']
                    hb = (hb * cf['n_large'] + h['hbm_bytes_per_launch_small'] * cf['n_small']) / max(cf['n_large'] + cf['n_small'], 1)
                if hb is not None:
                    b += hb * n
                    n_l += n
                for key in ('mfma_busy', 'valu_busy', 'lds_busy'):
                    if key in h:
                        busy.setdefault(key, []).append((h[key], n))
            per_grid = sorted(h['hbm_bytes_per_launch_large'] for h in hits if h.get('hbm_bytes_per_launch_large') is not None)
            sized = [h for h in hits if h.get('hbm_bytes_per_launch_small') is not None]
            if len(per_grid) > 1 and per_grid[-1] > 1.5 * per_grid[0]:      # launched with two grids = two map sizes
                e['hbm_bytes_per_launch_large'], e['hbm_bytes_per_launch_small'] = round(per_grid[-1]), round(per_grid[0])
                e['note'] = 'avg_us and the busy figures are means over BOTH sizes'
            elif sized:     # one kernel, one grid, two map sizes (image side / BEV side): the per-size byte counts as well
                cf = sized[0]['counters']['FETCH_SIZE']
                e['hbm_bytes_per_launch_large'] = round(sized[0]['hbm_bytes_per_launch_large'])
                e['hbm_bytes_per_launch_small'] = round(sized[0]['hbm_bytes_per_launch_small'])
                e['launch_share_large'] = round(cf['n_large'] / max(cf['n_large'] + cf['n_small'], 1), 3)
                e['note'] = 'avg_us and the busy figures are means over BOTH sizes'
            if n_l:
                e['hbm_bytes_per_launch'] = round(b / n_l)
                e['achieved_gbs'] = round(b / n_l / us / 1e3, 1)
                tot_bytes += b / n_l * calls
            for key, vs in busy.items():
                e[key] = round(sum(x * n for x, n in vs) / max(sum(n for _, n in vs), 1), 3)
        if calls >= 0.5 or us * calls > 2.0:
            kernels.append(e)
    out = dict(source=tag, kernel_stats=stats, pmc=pmc.get('source'), forwards_timed=n_fwd,
               formula='hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (separate rocprofv3 --pmc passes over the eager '
                       'forward); avg_us from rocprofv3 --kernel-trace of the serial forward (DI_OVERLAP=0, --inflight 1); busy '
                       'figures: SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES), SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES, '
                       'SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES',
               forward=dict(kernel_us=round(tot_us, 1), launches=round(sum(int(r['Calls']) for r in rows) / n_fwd, 1),
                            hbm_bytes_pmc=round(tot_bytes), hbm_rate_gbs=round(tot_bytes / tot_us / 1e3, 1),
                            frac_of_peak=round(tot_bytes / (tot_us * 1e-6) / HBM_PEAK, 4)),
               kernels=kernels)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

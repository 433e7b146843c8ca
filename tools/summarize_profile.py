"""Turn one gpurun_out/<tag>/ session (tools/gpu_session.sh) into the tracked summaries under profiles/.

    python tools/summarize_profile.py r01b

writes
    profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (copied as is)
    profiles/<tag>_pmc_hbm.csv          per kernel x grid: launches, FETCH_SIZE / WRITE_SIZE (KiB, raw averages
                                        per launch from the two separate --pmc passes) and the corrected HBM bytes
    profiles/pmc_local_attn.json        what bench.py reports as roofline.traffic for the dominant kernel

HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: rocprofv3 reports both in KiB, and on
gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) coalesced reads at 64 B
(/opt/skills/guides/MI355X_MICROARCH.md, "HBM"), so it is doubled; WRITE_SIZE needs no correction
(checked: the local-attention kernel's WRITE_SIZE equals its output tensor to the byte).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                agg[(r['Kernel_Name'], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return agg


def main(tag):
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    shutil.copy(os.path.join(src, 'prof', 'stats_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))
    fetch = per_kernel(os.path.join(src, 'pmc_fetch', 'pmc_counter_collection.csv'), 'FETCH_SIZE')
    write = per_kernel(os.path.join(src, 'pmc_write', 'pmc_counter_collection.csv'), 'WRITE_SIZE')
    rows = []
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        if 'di::' not in name:
            continue
        f = fetch.get(key, [])
        w = write.get(key, [])
        fa = sum(f) / len(f) if f else float('nan')
        wa = sum(w) / len(w) if w else float('nan')
        rows.append(dict(kernel=name.split('(')[0], grid_size=grid, launches=max(len(f), len(w)),
                         fetch_size_kib=round(fa, 2), write_size_kib=round(wa, 2),
                         hbm_bytes_per_launch=int(round((2 * fa + wa) * 1024))))
    with open(os.path.join(dst, f'{tag}_pmc_hbm.csv'), 'w', newline='') as f:
        wcsv = csv.DictWriter(f, fieldnames=list(rows[0]))
        wcsv.writeheader()
        wcsv.writerows(rows)
    # dominant kernel: the image-side launches of the local-window attention.  The persistent kernel uses
    # the same grid for the image-side (6x112x200) and BEV-side (180x180) maps, so the image-side
    # launches are told apart by their counter values (4x the bytes): keep values above half the maximum.
    la_keys = [key for key in set(fetch) | set(write) if 'local_attn' in key[0]]
    kname = max(la_keys, key=lambda key: max(fetch.get(key, [0])))[0]
    fvals = [x for key in la_keys if key[0] == kname for x in fetch.get(key, [])]
    wvals = [x for key in la_keys if key[0] == kname for x in write.get(key, [])]
    fbig = [x for x in fvals if x > 0.5 * max(fvals)]
    wbig = [x for x in wvals if x > 0.5 * max(wvals)]
    fa, wa = sum(fbig) / len(fbig), sum(wbig) / len(wbig)
    big = dict(kernel=kname.split('(')[0], grid_size=max(key[1] for key in la_keys if key[0] == kname),
               fetch_size_kib=round(fa, 2), write_size_kib=round(wa, 2),
               hbm_bytes_per_launch=int(round((2 * fa + wa) * 1024)), launches=len(fbig))
    with open(os.path.join(dst, 'pmc_local_attn.json'), 'w') as f:
        json.dump(dict(source=f'{tag}_pmc_hbm.csv', kernel=big['kernel'], grid_size=big['grid_size'],
                       fetch_size_kib=big['fetch_size_kib'], write_size_kib=big['write_size_kib'],
                       hbm_bytes_per_launch=big['hbm_bytes_per_launch'], image_side_launches=big['launches'],
                       formula='(2*FETCH_SIZE + WRITE_SIZE) * 1024'), f, indent=1)
    for r in rows:
        print(r)


if __name__ == '__main__':
    main(sys.argv[1])

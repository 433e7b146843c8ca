"""Per kernel (name, grid size, size class): mean of each counter over the dispatches of one rocprofv3 --pmc pass
(one JSON line per kernel), or --merge of the passes into the summary that profiles/pmc_*.json are cut from."""
import collections, csv, json, sys


def one(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].split('(')[0]
        agg[(name, r.get('Grid_Size', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
    for (name, grid), cs in agg.items():
        for c, v in cs.items():
            v = sorted(v)
            # launches of one kernel at two sizes (image side / BEV side) share a grid: split at the largest gap
            cut = max(range(1, len(v)), key=lambda i: v[i] / max(v[i - 1], 1e-9)) if len(v) > 3 else None
            big = v[cut:] if cut is not None and v[cut] > 1.8 * max(v[cut - 1], 1e-9) else v
            small = v[:cut] if big is not v else []
            print(json.dumps(dict(kernel=name, grid=grid, counter=c, n=len(v), mean=sum(v) / len(v),
                                  n_large=len(big), mean_large=sum(big) / len(big),
                                  n_small=len(small), mean_small=(sum(small) / len(small)) if small else None)))


def merge(path, tag):
    rows = [json.loads(l) for l in open(path)]
    res = collections.defaultdict(dict)
    for r in rows:
        res[f"{r['kernel']} grid={r['grid']}"][r['counter']] = {k: r[k] for k in ('n', 'mean', 'n_large', 'mean_large', 'n_small', 'mean_small')}
    out = {}
    for k, d in res.items():
        e = dict(counters=d)
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            for cls in ('large', 'small'):
                f, w = d['FETCH_SIZE'].get('mean_' + cls), d['WRITE_SIZE'].get('mean_' + cls)
                if f is not None and w is not None:
                    e[f'hbm_bytes_per_launch_{cls}'] = (2 * f + w) * 1024          # MI355X guide: FETCH_SIZE counts 2 KiB units on gfx950
        if 'SQ_BUSY_CU_CYCLES' in d:
            b = d['SQ_BUSY_CU_CYCLES']['mean']
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
                e['mfma_busy'] = d['SQ_VALU_MFMA_BUSY_CYCLES']['mean'] / (4 * b)
            if 'SQ_LDS_IDX_ACTIVE' in d:
                e['lds_busy'] = d['SQ_LDS_IDX_ACTIVE']['mean'] / b
            if 'SQ_ACTIVE_INST_VALU' in d:          # as DESIGN section 3 quotes it since round 2: VALU-active cycles / busy CU cycles
                e['valu_busy'] = d['SQ_ACTIVE_INST_VALU']['mean'] / b
        out[k] = e
    print(json.dumps(dict(source=tag, formula='(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, eager forward, one stream', kernels=out), indent=1))


if __name__ == '__main__':
    merge(sys.argv[2], sys.argv[3]) if sys.argv[1] == '--merge' else one(sys.argv[1])

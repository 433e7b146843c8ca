"""Per kernel (name x grid size): mean of every collected counter per dispatch, from the pN.csv files of a pmc session."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, 'p*.csv'))):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r['Kernel_Name'].split('(')[0][-60:]
            if 'sp::conv_kernel' in r['Kernel_Name']:
                name = r['Kernel_Name'].split('>')[0][-40:] + '>'
            key = (name, r.get('Grid_Size', ''))
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for key, cs in sorted(agg.items()):
    if not any(k in key[0] for k in ('conv3x3', 'local_attn', 'pointwise', 'i2p', 'mha_decode', 'tl_', 'dynconv', 'bevwarp', 'attn_dense', 'sp::conv', 'sp::nbr')):
        continue
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    res[f'{key[0]} grid={key[1]}'] = d
    print(key[0], 'grid', key[1])
    print('   ' + '  '.join(f'{c}={v:.4g}' for c, v in sorted(d.items())))
    if 'SQ_WAVE_CYCLES' in d and d['SQ_WAVE_CYCLES'] > 0:
        wc = d['SQ_WAVE_CYCLES']
        print('   fractions of wave-cycles: ' + '  '.join(f'{c[3:]}={d[c] / wc:.3f}' for c in
              ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_VALU') if c in d))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'SQ_BUSY_CU_CYCLES' in d and d['SQ_BUSY_CU_CYCLES'] > 0:
        print(f"   mfma_busy (MFMA busy cycles / (4 SIMD x busy CU cycles)) = {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * d['SQ_BUSY_CU_CYCLES']):.3f}")
    if 'SQ_LDS_IDX_ACTIVE' in d and 'SQ_BUSY_CU_CYCLES' in d and d['SQ_BUSY_CU_CYCLES'] > 0:
        print(f"   lds_busy (LDS index active / busy CU cycles) = {d['SQ_LDS_IDX_ACTIVE'] / d['SQ_BUSY_CU_CYCLES']:.3f}   bank conflict share = {d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
    if 'FETCH_SIZE' in d or 'WRITE_SIZE' in d:
        print(f"   HBM bytes per launch = 2*FETCH*1024 + WRITE*1024 = {(2 * d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) * 1024:.4g}")
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)

#!/bin/bash
# PMC passes over the image-side local attention (tools/la_bench.py <variant>); usage: bash tools/la_pmc.sh <tag> <variant>
TAG=$1; VAR=$2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
export TMPDIR=/tmp LA_ITERS=5
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $REPO/tools/la_bench.py $VAR > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/p*/pmc_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'local_attn' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f, k, len(v), sum(v) / len(v))
PY
find $OUT -name '*.csv' -size +5M -delete

#!/bin/bash
# rocprofv3 counter passes (separate runs, --kernel-trace only: gpurun refuses --pmc with other trace domains) over
# tools/kernel_pmc.py.  Usage: tools/pmc_session.sh TAG [kernel list]   (PMC_CMD="python tools/..." profiles another command)
TAG=${1:-pmc}; WHICH=${2:-conv,la,pw}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" \
         "GRBM_GUI_ACTIVE TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- ${PMC_CMD:-python $GRAFT_REPO_ROOT/tools/kernel_pmc.py $WHICH} > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/p$i.csv
  rm -rf $OUT/p$i
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT

#!/bin/bash
# Extra PMC passes (stall / queue counters) for the default bench workload.
TAG=${1:-x}
STEPS=${2:-30}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps $STEPS --warmup 5 --skip-cpu"
i=0
for C in "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LEVEL_WAVES" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/x$i -o pmc -- $CMD > $OUT/x$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -2 $OUT/x$i.log)"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/x*/pmc_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'stft_wave1024' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY

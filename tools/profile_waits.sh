#!/bin/bash
# Runs on the GPU box: one PMC pass of the bench kernel with the wave-state counters
# (active / waiting cycles by cause), summarised per dispatch.
TAG=${1:-waits}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 30 --warmup 5 --skip-cpu"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'stft_wave1024' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY

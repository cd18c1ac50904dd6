#!/bin/bash
# Runs on the GPU box: wave-state / instruction / LDS counters and the kernel trace of the two 1024 / 256 magnitude
# kernels (AA_W1024_X2=0: 8 x 8 x 8, one frame per wave at a time; 1: two frames per wave, 16 x 16 x 2), mean per dispatch.
TAG=${1:-x2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  export AA_W1024_X2=$V
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $C --output-format csv -d $OUT/v${V}_p$i -o pmc -- python3 $REPO/tools/run_geometry.py 1024 256 mag 60 > $OUT/v${V}_p$i.log 2>&1 || echo "x2=$V pass $i failed: $(tail -2 $OUT/v${V}_p$i.log)"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v${V}_kt -o kt -- python3 $REPO/tools/run_geometry.py 1024 256 mag 400 > $OUT/v${V}_kt.log 2>&1 || echo "x2=$V kt failed"
done
python3 - <<PY
import csv, glob, collections
for V in (0, 1):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob("$OUT/v%d_p*/**/*counter_collection.csv" % V, recursive=True)):
        for r in csv.DictReader(open(f)):
            if 'stft_wave1024' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                name = r['Kernel_Name']
    print("# AA_W1024_X2=%d  %s" % (V, name[:60]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
    for f in glob.glob("$OUT/v%d_kt/**/*kernel_stats.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stft_wave1024' in r['Name']:
                print("kernel-trace:", r['Name'][:50], "calls", r['Calls'], "avg ns", r['AverageNs'])
PY

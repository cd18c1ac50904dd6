#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the detector chains at scale (tools/run_detectors.py), and - in passes of
# their own, no tracing - the issue / wait counters of the tuner kernels.  Output: gpurun_out/r05_detectors/.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/r05_det
DST=$REPO/gpurun_out/r05_detectors
rm -rf $OUT; mkdir -p $OUT $DST
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/tools/run_detectors.py 20 > $OUT/kt.log 2>&1; echo "kt rc=$?"
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $DST/r05_detectors_rocprof_summary.csv
tail -3 $OUT/kt.log
if [ "$1" = "pmc" ]; then
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/tools/run_detectors.py 3 ${2:-tuner} > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed: $(tail -2 $OUT/pmc$i.log)"
    f=$(find $OUT/pmc$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 $REPO/tools/summarize_pmc.py "$f" >> $DST/r05_detectors_counters.txt
  done
fi
rm -rf $OUT

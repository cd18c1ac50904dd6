#!/bin/bash
# Runs on the GPU box: the L2 / memory-side counters of the complex-rows launch into a FAST and a SLOW output buffer of the
# same process (tools/probe_rows_contrast.py), one rocprofv3 --pmc pass per counter set, no tracing.  Output:
# gpurun_out/r05_rows_contrast.txt (per counter: the 30 fast and the 30 slow dispatches, summed and per instance).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/rows_contrast
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
touch $REPO/gpurun_out/r05_rows_contrast.txt
i=0
PASSES=${1:-1234}
for C in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_BUSY GRBM_GUI_ACTIVE" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL TCC_WRITE_REQ_LATENCY GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES TCC_REQ TCC_WRITE TCC_WRITEBACK GRBM_GUI_ACTIVE" \
         "TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_DRAM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  case $PASSES in *$i*) ;; *) continue;; esac
  echo "pass $i: $C"
  timeout -k 10 170 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/tools/probe_rows_contrast.py 160 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
  grep '^{' $OUT/p$i.log >> $REPO/gpurun_out/r05_rows_contrast.txt
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $REPO/tools/summarize_contrast.py "$f" >> $REPO/gpurun_out/r05_rows_contrast.txt
  [ $i = 1 ] && [ -n "$f" ] && head -3 "$f" > $REPO/gpurun_out/r05_rows_contrast_csv_head.txt
done
rm -rf $OUT

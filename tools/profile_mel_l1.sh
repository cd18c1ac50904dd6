#!/bin/bash
# Runs on the GPU box (via gpurun): the L1 / texture path of the 2048 log-mel kernel with the filter
# weights streamed from the per-lane table (shipped) and rebuilt on the VALU (AA_MEL_NO_RESIDENT=1).
# Output: gpurun_out/r02_profiles/r02_mel_l1_counters.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/mel_l1
rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out/r02_profiles
cd /tmp && export TMPDIR=/tmp
for V in streamed arith; do
  E=0; [ $V = arith ] && E=1
  for P in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    T=$(echo $P | cut -d' ' -f1)
    AA_MEL_NO_RESIDENT=$E rocprofv3 --pmc $P --output-format csv -d $OUT/${V}_$T -o pmc -- python3 $REPO/tools/run_geometry.py 2048 512 logmel 30 2 > $OUT/${V}_$T.log 2>&1 || echo "$V $T failed: $(tail -1 $OUT/${V}_$T.log)"
  done
done
python3 - <<PY
import csv, glob, collections
out = open("$REPO/gpurun_out/r02_profiles/r02_mel_l1_counters.txt", "w")
out.write("# tools/profile_mel_l1.sh: stft_wave2048_kernel log-mel (2048 / 512, 128 bands, two inputs alternating), mean per dispatch;\\n")
out.write("# streamed = filter weights from the per-lane table (shipped), arith = rebuilt from the triangle edges (AA_MEL_NO_RESIDENT=1)\\n")
for v in ("streamed", "arith"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s_*/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "stft_wave2048" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    out.write("== %s\\n" % v)
    for c in sorted(acc):
        out.write("  %-32s n=%4d mean=%.6g\\n" % (c, len(acc[c]), sum(acc[c]) / len(acc[c])))
out.close()
print(open("$REPO/gpurun_out/r02_profiles/r02_mel_l1_counters.txt").read())
PY

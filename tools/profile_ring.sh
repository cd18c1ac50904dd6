#!/bin/bash
# Runs on the GPU box (via gpurun): wave-state counters of the 2048 kernels with the two chunk buffers
# (library built with -DAA_EXPERIMENTS=1; AA_W2048_RING=0) and with the ring two chunks ahead, two inputs alternating (samples from HBM).
# Output: gpurun_out/r02_profiles/r02_ring_wait_counters.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/ring
rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out/r02_profiles
cd /tmp && export TMPDIR=/tmp
for K in logmel mag; do
  for R in 0 1; do
    AA_W2048_RING=$R rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${K}_ring$R -o pmc -- python3 $REPO/tools/run_geometry.py 2048 512 $K 40 2 > $OUT/${K}_ring$R.log 2>&1 || echo "$K ring=$R failed: $(tail -2 $OUT/${K}_ring$R.log)"
    AA_W2048_RING=$R rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${K}_kt$R -o kt -- python3 $REPO/tools/run_geometry.py 2048 512 $K 200 2 > $OUT/${K}_kt$R.log 2>&1 || echo "$K kt ring=$R failed"
  done
done
python3 - <<PY
import csv, glob, collections, os
out = open("$REPO/gpurun_out/r02_profiles/r02_ring_wait_counters.txt", "w")
out.write("# tools/profile_ring.sh: stft_wave2048_kernel, 2048 / 512, two inputs alternating (samples from HBM);\n")
out.write("# ring0 = two chunk buffers, DMA one chunk ahead; ring1 = ring of 30 blocks, DMA two chunks ahead.\n")
out.write("# PMC: mean per dispatch; kernel trace: average duration over 200 dispatches.\n")
for k in ("logmel", "mag"):
    for r in (0, 1):
        acc = collections.defaultdict(list)
        for f in glob.glob("$OUT/%s_ring%d/**/*counter_collection.csv" % (k, r), recursive=True):
            for row in csv.DictReader(open(f)):
                if "stft_wave2048" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur = []
        for f in glob.glob("$OUT/%s_kt%d/**/*kernel_stats.csv" % (k, r), recursive=True):
            for row in csv.DictReader(open(f)):
                if "stft_wave2048" in row["Name"]:
                    dur.append((row["Calls"], row["AverageNs"]))
        out.write("== %s ring%d  kernel trace (calls, avg ns): %s\n" % (k, r, dur))
        for c in sorted(acc):
            out.write("  %-24s n=%4d mean=%.6g\n" % (c, len(acc[c]), sum(acc[c]) / len(acc[c])))
out.close()
print(open("$REPO/gpurun_out/r02_profiles/r02_ring_wait_counters.txt").read())
PY

#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the log-mel / MFCC geometries
# (fused path and AA_MEL_GEMM=1), plus one PMC pass with the MFMA counters for the GEMM kernels.
TAG=${1:-r01_mel}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $OUT/counters.txt | sort -u > $OUT/mfma_counters.txt
cat $OUT/mfma_counters.txt
export AA_MEL_GEMM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_fused -o kt -- python3 $REPO/tools/mel_trace.py > $OUT/kt_fused.log 2>&1
export AA_MEL_GEMM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_gemm -o kt -- python3 $REPO/tools/mel_trace.py > $OUT/kt_gemm.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_gemm -o pmc -- python3 $REPO/tools/mel_trace.py > $OUT/pmc_gemm.log 2>&1 || tail -3 $OUT/pmc_gemm.log
export AA_MEL_GEMM=0
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_fused -o pmc -- python3 $REPO/tools/mel_trace.py > $OUT/pmc_fused.log 2>&1 || tail -3 $OUT/pmc_fused.log
python3 - <<PY
import csv, glob, collections
for mode in ("fused", "gemm"):
    for f in glob.glob("$OUT/kt_%s/**/*kernel_stats.csv" % mode, recursive=True):
        print("==", mode, "kernel stats")
        for r in csv.DictReader(open(f)):
            print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_ns {float(r['AverageNs']):10.0f} pct {r['Percentage']}")
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % mode, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        print("==", mode, "pmc (mean per dispatch)")
        for k, d in acc.items():
            print(" ", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY

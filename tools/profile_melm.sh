#!/bin/bash
# Runs on the GPU box: PMC counters of the 1024 / 256 log-mel and MFCC launches, for each library named on the command line
# (name=path[@ENV=VALUE]).  Output: gpurun_out/prof/melm/<name>_<kind>_<pass>/ + a summary on stdout.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/melm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SPEC in "$@"; do
  NAME=${SPEC%%=*}; REST=${SPEC#*=}
  LIB=${REST%%@*}
  ENVS=""; [ "$REST" != "$LIB" ] && ENVS=${REST#*@}
  for KIND in ${KINDS:-logmel mfcc}; do
    i=0
    for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      D=$OUT/${NAME}_${KIND}_$i
      if [ -n "$ENVS" ]; then export ${ENVS//@/ }; fi
      AA_TOOL_LIB=$REPO/$LIB rocprofv3 --pmc $C --output-format csv -d $D -o pmc -- python3 $REPO/tools/run_geometry.py ${NFFT:-1024} ${HOP:-256} $KIND 30 > $D.log 2>&1 || echo "$NAME $KIND pass $i failed: $(tail -2 $D.log)"
      if [ -n "$ENVS" ]; then for kv in ${ENVS//@/ }; do unset ${kv%%=*}; done; fi
    done
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_[0-9]")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stft_wave" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("#", os.path.basename(d))
    for k in sorted(acc):
        v = acc[k]
        print(f"  {k:32s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
PY

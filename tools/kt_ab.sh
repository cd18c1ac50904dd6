#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of one tool script under two library builds (A/B of kernel
# times without the host's share).  usage: tools/kt_ab.sh <script.py> <libA.so> [<libB.so> ...]; output: gpurun_out/kt_ab_<lib>.csv
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SCRIPT=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  OUT=/tmp/kt_ab_$N
  rm -rf $OUT
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/$SCRIPT $REPO/$L > /tmp/kt_ab_$N.log 2>&1; echo "$N rc=$?"
  f=$(find $OUT -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $REPO/gpurun_out/kt_ab_$N.csv && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    if float(r["Percentage"]) > 1.0:
        print(f"   {n:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.2f} us")
PY
done

#!/bin/bash
# Runs on the GPU box: address-translation counters of one geometry / output kind in several fresh processes (the
# complex 1024 / 256 launch is fast or slow per PROCESS, profiles/r02_bimodal_complex.txt): is the slow mode a
# translation (TLB) effect of where the driver put the output buffer?   tools/probe_tlb.sh [n_fft hop kind procs]
N=${1:-1024}; HOP=${2:-256}; KIND=${3:-complex}; PROCS=${4:-6}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/tlb_${N}_${KIND}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $PROCS); do
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/tools/run_geometry.py $N $HOP $KIND 80 > $OUT/p$i.log 2>&1 || echo "proc $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 - <<PY
import csv, glob, collections
for i in range(1, $PROCS + 1):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stft_wave' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    m = {k: sum(v[20:]) / max(len(v[20:]), 1) for k, v in acc.items()}
    print("proc %d: " % i + "  ".join("%s=%.4g" % (k.replace('TCP_UTCL1_', 'L1_').replace('_sum', ''), v) for k, v in sorted(m.items())))
PY

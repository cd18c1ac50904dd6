#!/bin/bash
# Runs on the GPU box: L2 -> memory (EA) write-path counters of the complex 1024 / 256 launch in several fresh processes (the launch is
# fast or slow per PROCESS, profiles/r02_bimodal_complex.txt), two counter sets alternating; GRBM_GUI_ACTIVE in each gives the mode.
PROCS=${1:-8}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/ea
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $PROCS); do
  if [ $((i % 2)) = 1 ]; then C="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE"
  else C="TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE"; fi
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/tools/run_geometry.py 1024 256 complex 60 > $OUT/p$i.log 2>&1 || echo "proc $i failed: $(tail -2 $OUT/p$i.log)"
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
    print("proc %d: " % i + "  ".join("%s=%.4g" % (k.replace('TCC_EA0_', '').replace('_sum', ''), v) for k, v in sorted(m.items())))
PY

REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/gridprobe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/p$i -o kt -- python3 $REPO/bench.py --skip-cpu --steps 50 --sustained-s 0.05 --config-steps 100 > $OUT/p$i.log 2>&1
  python3 - <<PY
import csv,glob,collections,json
rows=[]
for f in glob.glob("$OUT/p$i/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mel_kernel" in r["Kernel_Name"] or "stft_wave1024_kernel<(aa::OutKind)1" in r["Kernel_Name"] or "wave1024_kernelILi1" in r["Kernel_Name"]:
            rows.append(r)
by=collections.defaultdict(list)
for r in rows:
    key=(r["Kernel_Name"][:60], r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Scratch_Size"))
    by[key].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
line=json.loads([l for l in open("$OUT/p$i.log") if l.startswith('{"metric"')][-1])
c=line["configs"]
print("process $i bench: mfcc %.1f logmel %.1f" % (c["cfg4_mfcc_1024"]["ms"]*1e3, c["logmel_1024_256"]["ms"]*1e3))
for k,v in by.items():
    v=sorted(v)
    if len(v)>20: print("   ",k,"n",len(v),"median %.1f us"%(v[len(v)//2]/1e3),"p90 %.1f"%(v[int(len(v)*0.9)]/1e3))
PY
done
rm -rf $OUT

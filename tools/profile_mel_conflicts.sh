REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/melpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in "1024 256 logmel" "1024 256 mfcc" "2048 512 logmel"; do
  T=$(echo $G | tr ' ' '_')
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/mel_$T -o pmc -- python3 $REPO/tools/run_geometry.py $G 30 > $OUT/mel_$T.log 2>&1 || echo "mel $T failed: $(tail -2 $OUT/mel_$T.log)"
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/mel_$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mel_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in acc.items()}
print("$T", m, "ratio %.4f" % (m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"]))
PY
done
rm -rf $OUT

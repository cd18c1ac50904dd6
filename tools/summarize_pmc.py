"""Condenses a rocprofv3 counter_collection.csv: per kernel name and counter, the mean over dispatches (summed over the
dimensions a dispatch reports), and the dispatch count."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
for k in sorted(acc):
    n = len(disp[k])
    print(f"{k}  [{n} dispatches]")
    for c in sorted(acc[k]):
        print(f"    {c:28s} {acc[k][c] / n:16.1f}")

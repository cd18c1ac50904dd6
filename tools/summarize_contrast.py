"""rocprofv3 counter_collection.csv of tools/probe_rows_contrast.py: the last 60 dispatches of the stft kernel are 30 launches
into the fast buffer, then 30 into the slow one.  Per counter: the per-dispatch total (all instances) for both, and - where the
file reports instances separately - the spread over the instances (min / mean / max, and how unevenly the work is spread)."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))       # counter -> dispatch -> values (one per row)
order = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "stft_wave" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        rows[r["Counter_Name"]][d].append(float(r["Counter_Value"]))
        order.append(d)
disp = sorted(set(order))[-60:]
fast, slow = disp[:30], disp[30:]
for c in sorted(rows):
    def stats(ds):
        tot = [sum(rows[c][d]) for d in ds]
        inst = [rows[c][d] for d in ds]
        n_inst = len(inst[0])
        mean_tot = sum(tot) / len(tot)
        if n_inst > 1:
            per = [sum(v[k] for v in inst) / len(inst) for k in range(n_inst)]
            return mean_tot, n_inst, min(per), max(per)
        return mean_tot, 1, mean_tot, mean_tot
    f, s = stats(fast), stats(slow)
    print(f"  {c:36s} fast {f[0]:14.4g} [{f[1]} inst: {f[2]:.4g} .. {f[3]:.4g}]   slow {s[0]:14.4g} [{s[1]} inst: {s[2]:.4g} .. {s[3]:.4g}]   slow/fast {s[0] / f[0] if f[0] else float('nan'):.3f}")

"""Condenses the rocprofv3 output of tools/profile_r05.sh into profiles/r05_*: per-kernel average
durations from the kernel traces, HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes
(gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts half the bytes of 16-B-per-lane
streaming reads -> doubled; WRITE_SIZE as is), wave-state counters, LDS bank conflicts.
Usage: python3 tools/summarize_r05.py gpurun_out/prof/r05"""
import collections
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "gpurun_out", "r05_profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"(stft_\w+_kernel|rows_gemm_mfma_kernel|downmix_kernel|\w+_kernel)<?([^>]*)", name)
    if not m:
        return name[:60]
    args = re.findall(r"-?\d+|true|false", m.group(2))
    return m.group(1) + ("<" + ",".join(args[:4]) + ">" if args else "")


def kernel_stats(tag):
    rows = []
    for f in glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    return rows


for tag in ("kt_head", "kt_all"):
    rows = kernel_stats(tag)
    with open(os.path.join(dst, f"r05_{tag}_rocprof_summary.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats of tools/profile_r05.sh (%s)\n" % tag)
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        for r in rows:
            # (kernel names hold commas - template arguments: quoted)
            f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                 r["MinNs"], r["MaxNs"], r["Percentage"]))


def counters(pattern, kernel_filter):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_filter in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


lines = []
head = counters("pmc[1-5]", "stft_wave1024_kernel")
traffic = {}
for k, cs in sorted(head.items(), key=lambda kv: kv[0].startswith("stft_wave1024_kernel<1,")):   # (the magnitude variant last: it is the headline)
    lines.append(f"# {k}  (python3 bench.py --steps 30 --warmup 5 --skip-cpu --skip-configs, mean per dispatch)")
    for c, v in sorted(cs.items()):
        lines.append(f"{c:28s} n={len(v):4d} mean={sum(v) / len(v):.6g}")
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fs, ws = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        traffic = {"round": 5, "kernel": k,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_r05.sh), mean per "
                             "dispatch, KB = 1024 B; gfx950 correction: FETCH_SIZE counts 1/2 of wide (16 B/lane) coalesced "
                             "reads -> doubled; WRITE_SIZE taken as is",
                   "fetch_size_kb": fs, "write_size_kb": ws,
                   "hbm_read_bytes_per_launch": 2 * fs * 1024, "hbm_write_bytes_per_launch": ws * 1024,
                   "hbm_bytes_per_launch": 2 * fs * 1024 + ws * 1024,
                   "algorithmic_bytes_per_launch": 3076 * 112497}
open(os.path.join(dst, "r05_wave_state_counters.txt"), "w").write("\n".join(lines) + "\n")
if traffic:
    json.dump(traffic, open(os.path.join(dst, "r05_pmc_traffic.json"), "w"), indent=1)

allc = counters("pmc[6-9]", "")
lines = ["# every kernel of python3 bench.py --skip-cpu (configs object), mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB"]
for k, cs in sorted(allc.items()):
    lines.append(f"# {k}")
    for c, v in sorted(cs.items()):
        lines.append(f"  {c:30s} n={len(v):4d} mean={sum(v) / len(v):.6g}")
open(os.path.join(dst, "r05_all_kernels_counters.txt"), "w").write("\n".join(lines) + "\n")

lines = ["# the fused mel / MFCC kernels of r4 (band-blocked contraction on v_mfma_f32_4x4x1), python3 tools/run_geometry.py <n_fft> <hop>",
         "# <kind> 30, mean per dispatch.  conflict ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (r3: 0.34 with the gather)"]
for g in ("1024_256_logmel", "1024_256_mfcc", "2048_512_logmel"):
    cs = counters("mel_" + g, "mel_kernel")
    for k, d in cs.items():
        lines.append(f"# {g}: {k}")
        for c, vals in sorted(d.items()):
            lines.append(f"  {c:28s} n={len(vals):4d} mean={sum(vals) / len(vals):.6g}")
        if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
            m = lambda n: sum(d[n]) / len(d[n])
            lines.append(f"  conflict ratio               {m('SQ_LDS_BANK_CONFLICT') / m('SQ_LDS_IDX_ACTIVE'):.4f}")
open(os.path.join(dst, "r05_mel_kernels_lds_counters.txt"), "w").write("\n".join(lines) + "\n")
for f in sorted(os.listdir(dst)):
    print("==", f)
    print(open(os.path.join(dst, f)).read()[:3000])

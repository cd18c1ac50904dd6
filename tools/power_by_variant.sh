#!/bin/bash
# Runs on the GPU box: package power and sclk (rocm-smi, 4 samples a second) while tools/ablate_w1024 holds
# each ablation variant of the 1024 kernel for a few seconds; prints the mean of the samples inside each
# variant's window.  Usage: bash tools/power_by_variant.sh [seconds per variant]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SEC=${1:-4}
LOG=/tmp/power_samples.txt
: > $LOG
( while true; do
    T=$(date +%s.%N)
    L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' ')
    echo "$T $L" >> $LOG
    sleep 0.2
  done ) &
SAMPLER=$!
AA_ABL_SUSTAIN=$SEC $REPO/tools/bin/ablate_w1024 8 | grep SUSTAIN > /tmp/sustain.txt
kill $SAMPLER
python3 - <<'PY'
import re
samples = []
for line in open("/tmp/power_samples.txt"):
    t = float(line.split()[0])
    p = re.search(r"Package Power \(W\): ([0-9.]+)", line)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line)
    if p and c:
        samples.append((t, float(p.group(1)), int(c.group(1))))
for line in open("/tmp/sustain.txt"):
    m = re.match(r"SUSTAIN (.{44}) begin ([0-9.]+) end ([0-9.]+)  last ([0-9.]+) us/launch", line)
    name, b, e, us = m.group(1).strip(), float(m.group(2)), float(m.group(3)), float(m.group(4))
    win = [(p, c) for t, p, c in samples if b + 1.5 < t < e]      # skip the first 1.5 s of each window
    if win:
        print(f"{name:44s} {us:7.2f} us/launch   {sum(p for p, _ in win) / len(win):7.1f} W   sclk {sum(c for _, c in win) / len(win):6.0f} MHz   ({len(win)} samples)")
    else:
        print(f"{name:44s} {us:7.2f} us/launch   no samples")
PY

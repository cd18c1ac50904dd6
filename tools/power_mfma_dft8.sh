#!/bin/bash
# Runs on the GPU box: package power and sclk (rocm-smi, 5 samples a second) while tools/bin/ubench_mfma_dft8
# holds each variant of one radix-8 pass; energy per pass = (package W - idle W) x time per pass.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SEC=${1:-5}
LOG=/tmp/power_samples.txt
: > $LOG
sleep 1
IDLE=$(rocm-smi --showpower 2>/dev/null | grep -oE "Package Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
( while true; do
    T=$(date +%s.%N)
    L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' ')
    echo "$T $L" >> $LOG
    sleep 0.2
  done ) &
SAMPLER=$!
$REPO/tools/bin/ubench_mfma_dft8 $SEC | grep SUSTAIN > /tmp/sustain.txt
kill $SAMPLER
IDLE=$IDLE python3 - <<'PY'
import os, re
idle = float(os.environ.get("IDLE") or 236.0)
samples = []
for line in open("/tmp/power_samples.txt"):
    t = float(line.split()[0])
    p = re.search(r"Package Power \(W\): ([0-9.]+)", line)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line)
    if p and c:
        samples.append((t, float(p.group(1)), int(c.group(1))))
print(f"# idle package power before the run: {idle:.0f} W")
for line in open("/tmp/sustain.txt"):
    m = re.match(r"SUSTAIN (.{44}) begin ([0-9.]+) end ([0-9.]+)  last ([0-9.]+) us/launch  ([0-9.]+) ns/pass/wave-slot  passes ([0-9.e+]+)", line)
    name, b, e, us, nsp, passes = m.group(1).strip(), float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))
    win = [(p, c) for t, p, c in samples if b + 1.5 < t < e]
    if win:
        pw = sum(p for p, _ in win) / len(win)
        ck = sum(c for _, c in win) / len(win)
        nj = (pw - idle) * us * 1e-6 / passes * 1e9
        print(f"{name:44s} {us:9.1f} us/launch  {nsp:7.3f} ns per pass per wave slot  {pw:7.1f} W  sclk {ck:5.0f} MHz  {nj:7.3f} nJ per pass (dynamic)")
    else:
        print(f"{name:44s} {us:9.1f} us/launch  no samples")
PY

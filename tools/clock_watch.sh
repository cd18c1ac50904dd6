#!/bin/bash
# Runs on the GPU box: samples rocm-smi clocks / power while the bench kernel runs back to back.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $REPO/tools/clock_drift.py > /tmp/drift.txt 2>&1 &
PID=$!
sleep 9        # torch import + corpus synthesis, then the launches run for ~12 s
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Package Power" | tr '\n' ' '
  echo
  sleep 0.5
done
wait $PID
tail -4 /tmp/drift.txt
echo "idle:"; sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Package Power" | tr '\n' ' '; echo

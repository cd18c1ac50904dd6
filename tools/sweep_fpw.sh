#!/bin/bash
# A/B sweep of frames-per-workgroup (and kernel variant) on the default bench workload.
for f in ${FPWS:-4 8}; do
  AA_W1024_FRAMES_PER_WG=$f timeout -k 10 120 python bench.py --skip-cpu --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fpw $f', '%.4g frames/s' % d['value'], '%.4f ms' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])"
done

"""Reads the per-workgroup clock records tools/ablate_w1024 writes with AA_CLK_DUMP=<prefix> and
prints when the workgroups of each dispatch round (blockIdx / 256) finish.
Usage: python tools/wg_finish_order.py gpurun_out/clk_0_4.bin [more.bin ...]   (diagnostic)"""
import sys

import numpy as np

for name in sys.argv[1:]:
    a = np.fromfile(name, dtype=np.uint64).reshape(-1, 4)
    life = a[:, 1] * 0.01
    start = (a[:, 2] - a[:, 2].min()) * 0.01
    end = start + life
    b = np.arange(len(a))
    print(f"{name}: {len(a)} workgroups, kernel end {end.max():.1f} us, mean end {end.mean():.1f} us")
    for s in range((len(a) + 255) // 256):
        m = (b // 256) == s
        print(f"   blockIdx {s * 256:4d}..{s * 256 + 255:4d}: start {start[m].mean():5.2f}  end mean {end[m].mean():5.1f}"
              f"  min {end[m].min():5.1f}  max {end[m].max():5.1f} us")
    xcc = (a[:, 3] >> np.uint64(32)).astype(int)
    hw = (a[:, 3] & np.uint64(0xffffffff)).astype(int)
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    for x in sorted(set(xcc)):
        m = xcc == x
        print(f"   XCC {x}: {m.sum():4d} workgroups  end mean {end[m].mean():5.1f}  min {end[m].min():5.1f}  max {end[m].max():5.1f} us")
    key = xcc * 4096 + se * 256 + sh * 16 + cu
    cu_end = {}
    for k, e in zip(key, end):
        cu_end[k] = max(cu_end.get(k, 0.0), e)
    ce = np.array(sorted(cu_end.values()))
    print(f"   per CU (last resident): {len(ce)} CUs  end p10 {ce[len(ce)//10]:.1f}  p50 {ce[len(ce)//2]:.1f}  p90 {ce[len(ce)*9//10]:.1f}  max {ce[-1]:.1f}")

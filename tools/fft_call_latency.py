"""Per-call latency of the FftProcessor-shaped host entry points (aa_fft_forward / aa_fft_inverse):
one frame in pageable host memory in, one spectrum out, synchronous - the reference's inner seam
(src/dsp/fft.rs:33-41) used literally.  Usage: python tools/fft_call_latency.py   (diagnostic)"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_analyzer_rs_amd as aa  # noqa: E402

for n in (256, 1024, 2048, 4096):
    p = aa.FftProcessor(n)
    x = np.random.default_rng(0).standard_normal(n).astype(np.float32)
    for _ in range(200):
        p.process_forward(x)
    t0 = time.perf_counter()
    reps = 2000
    for _ in range(reps):
        spec = p.process_forward(x)
    t1 = time.perf_counter()
    for _ in range(200):
        p.process_inverse(spec)
    t2 = time.perf_counter()
    for _ in range(reps):
        p.process_inverse(spec)
    t3 = time.perf_counter()
    print(f"len {n:5d}: forward {1e6 * (t1 - t0) / reps:7.1f} us/call   inverse {1e6 * (t3 - t2) / reps:7.1f} us/call")

"""Tuner floor launch time (HIP events) over stream counts for the 10-minute 2048/512 rows; optional library path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import audio_analyzer_rs_amd as aa

if len(sys.argv) > 1:
    aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
mags = aa.Stft(2048, 512, output="mag")(xd)
nf, nb = mags.shape
gf = torch.full((nf,), 0.5, device="cuda")
name = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "libaa_stft.so"
for ns in (64, 128, 192, 256, 384, 512, 1024, 2048, 4096):
    offs = torch.tensor(np.linspace(0, nf, ns + 1).astype(np.int64), device="cuda")
    fn = lambda: aa.tuner_floor(mags, gf, offsets=offs)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name} {ns:5d} streams: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)

"""Launch times (HIP events, 20 launches) of the tuner floor and the pitch extraction on the 10-minute stream; an optional
library path as the first argument (A/B builds)."""
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
offs = torch.tensor(np.linspace(0, nf, 257).astype(np.int64), device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


eff, _ = aa.tuner_floor(mags, gf, offsets=offs)
name = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "libaa_stft.so"
print(name, "tuner_floor (256 streams) %.1f us" % timed(lambda: aa.tuner_floor(mags, gf, offsets=offs)),
      " tuner_pitches %.1f us" % timed(lambda: aa.tuner_pitches(mags, eff, 48000 / 2048)),
      " (24..4200 Hz: %.1f us)" % timed(lambda: aa.tuner_pitches(mags, eff, 48000 / 2048, 24.0, 4200.0)), flush=True)
f, s, c = aa.tuner_pitches(mags, eff, 48000 / 2048)
print("  checksum", float(f.double().sum()), float(s.double().sum()), int(c.sum()))

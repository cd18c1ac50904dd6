"""Log-mel launch time over band counts for both fused geometries (10-minute stream): the band-blocked contraction's
plan (quads, K-slices, steps) depends on the band count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_analyzer_rs_amd as aa

if len(sys.argv) > 1:
    aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n, hop in ((1024, 256), (2048, 512)):
    mag = aa.Stft(n, hop, output="mag")
    om = mag(xd)
    base = timed(lambda: mag.execute_device(xd, out=om))
    row = [f"{n}/{hop}: mag {base:.1f} us |"]
    for nm in (24, 40, 64, 80, 96, 128):
        plan = aa.Stft(n, hop, output="logmel", sample_rate=48000, n_mels=nm)
        out = plan(xd)
        row.append(f"{nm}: {timed(lambda: plan.execute_device(xd, out=out)):.1f} ({plan.mel_path})")
    print(" ".join(row), flush=True)

"""Times the 2048 / 512 MFCC and log-mel plans on the 10-minute stream (launch time from HIP events over 50 launches)
and prints the plan's paths - the tuner geometry's MFCC is the one configuration that is not a single launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_analyzer_rs_amd as aa

if len(sys.argv) > 1:
    aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])

xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()


def timed(fn, reps=50):
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


for n, hop in ((2048, 512), (1024, 256)):
    for kind, kw in (("logmel", dict(n_mels=64)), ("mfcc", dict(n_mels=64, n_mfcc=13)), ("mfcc", dict(n_mels=64, n_mfcc=40)),
                     ("mfcc", dict(n_mels=128, n_mfcc=20))):
        plan = aa.Stft(n, hop, output=kind, sample_rate=48000, **kw)
        out = plan(xd)
        us = timed(lambda: plan.execute_device(xd, out=out))
        print(f"{n}/{hop} {kind:6s} {kw}: {plan.kernel} / {plan.mel_path}  {us:7.1f} us  rows {tuple(out.shape)}", flush=True)

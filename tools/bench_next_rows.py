"""Times the "next"-row kernels (onset flux, onset gates, tuner noise floor) on one 10-minute
stream: these are recurrences over frames (one wave / one lane per bin), listed for
completeness — their parallelism is across streams, not within one."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import audio_analyzer_rs_amd as aa

x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


mags = aa.Stft(256, 64, output="mag")(xd)
nf, nb = mags.shape
gf = torch.full((nf,), float(10.0 ** (-60 / 20.0) * nb / 2.0), device="cuda")
flux, energy = aa.onset_flux(mags)
print(f"onset geometry 256/64: {nf} frames x {nb} bins")
print(f"  onset_flux   {timed(lambda: aa.onset_flux(mags)) * 1e3:9.3f} ms")
print(f"  onset_gates  {timed(lambda: aa.onset_gates(mags, flux, energy, gf)) * 1e3:9.3f} ms")
del mags
mags = aa.Stft(2048, 512, output="mag")(xd)
nf, nb = mags.shape
gf = torch.full((nf,), float(10.0 ** (-60 / 20.0) * nb / 2.0), device="cuda")
print(f"tuner geometry 2048/512: {nf} frames x {nb} bins")
print(f"  tuner_floor  {timed(lambda: aa.tuner_floor(mags, gf)) * 1e3:9.3f} ms")
eff, _ = aa.tuner_floor(mags, gf)
print(f"  tuner_pitches {timed(lambda: aa.tuner_pitches(mags, eff, 48000 / 2048)) * 1e3:8.3f} ms")

# the same frames as a batch of independent streams (the recurrences parallelise across streams)
mags256 = aa.Stft(256, 64, output="mag")(xd)
nf, nb = mags256.shape
gf = torch.full((nf,), float(10.0 ** (-60 / 20.0) * nb / 2.0), device="cuda")
for ns in (1, 64, 256, 1024, 4096, 16384):
    offs = torch.tensor(np.linspace(0, nf, ns + 1).astype(np.int64), device="cuda")
    flux, energy = aa.onset_flux(mags256, offsets=offs)
    df = timed(lambda: aa.onset_flux(mags256, offsets=offs), reps=5)
    dt = timed(lambda: aa.onset_gates(mags256, flux, energy, gf, offsets=offs), reps=2)
    print(f"{nf} frames as {ns:5d} streams: onset_flux {df * 1e3:9.3f} ms, onset_gates {dt * 1e3:9.3f} ms")
for ns in (1, 16, 256):
    offs = torch.tensor(np.linspace(0, mags.shape[0], ns + 1).astype(np.int64), device="cuda")
    gft = torch.full((mags.shape[0],), 0.5, device="cuda")
    dt = timed(lambda: aa.tuner_floor(mags, gft, offsets=offs), reps=2)
    print(f"tuner_floor, {mags.shape[0]} frames as {ns:5d} streams: {dt * 1e3:9.3f} ms")

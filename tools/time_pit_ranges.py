import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import audio_analyzer_rs_amd as aa
if len(sys.argv) > 1:
    aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
mags = aa.Stft(2048, 512, output="mag")(xd)
nf, nb = mags.shape
gf = torch.full((nf,), 0.5, device="cuda")
offs = torch.tensor(np.linspace(0, nf, 257).astype(np.int64), device="cuda")
eff, _ = aa.tuner_floor(mags, gf, offsets=offs)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
name = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "libaa_stft.so"
print(name, " ".join("%g Hz: %.1f us" % (mx, timed(lambda: aa.tuner_pitches(mags, eff, 48000 / 2048, 24.0, mx))) for mx in (10000.0, 4200.0, 2000.0, 1000.0)), flush=True)

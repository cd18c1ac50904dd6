"""Runs both detector chains at scale a few times on the 10-min corpus (for rocprofv3 passes): onset flux + gates on the
256/64 rows as 1024 streams and as ONE stream, tuner floor (256 streams) + pitch extraction on the 2048/512 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audio_analyzer_rs_amd as aa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
what = sys.argv[2] if len(sys.argv) > 2 else "all"
xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
if what in ("all", "onset"):
    mags = aa.Stft(256, 64, output="mag")(xd)
    nf, nb = mags.shape
    gf = torch.full((nf,), float(10.0 ** (-60 / 20.0) * nb / 2.0), device="cuda")
    offs = torch.tensor(np.linspace(0, nf, 1025).astype(np.int64), device="cuda")
    for _ in range(reps):
        f, e = aa.onset_flux(mags, offsets=offs)
        g, _ = aa.onset_gates(mags, f, e, gf, offsets=offs)
    f, e = aa.onset_flux(mags)
    g1, _ = aa.onset_gates(mags, f, e, gf)
    torch.cuda.synchronize()
    print("onset events (1024 streams / one stream)", int((g["flags"] & 16).bool().sum()), int((g1["flags"] & 16).bool().sum()))
    del mags
if what in ("all", "tuner"):
    mags = aa.Stft(2048, 512, output="mag")(xd)
    nf, nb = mags.shape
    gf = torch.full((nf,), 0.5, device="cuda")
    offs = torch.tensor(np.linspace(0, nf, 257).astype(np.int64), device="cuda")
    for _ in range(reps):
        eff, _ = aa.tuner_floor(mags, gf, offsets=offs)
        f, s, c = aa.tuner_pitches(mags, eff, 48000 / 2048)
    torch.cuda.synchronize()
    print("mean notes per frame", float(c.float().mean()))

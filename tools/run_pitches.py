"""Runs the tuner chain (noise floor as 256 streams + pitch extraction) a few times on the 10-min corpus (for rocprofv3 passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audio_analyzer_rs_amd as aa
xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
mags = aa.Stft(2048, 512, output="mag")(xd)
nf, nb = mags.shape
gf = torch.full((nf,), 0.5, device="cuda")
offs = torch.tensor(np.linspace(0, nf, 257).astype(np.int64), device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    eff, _ = aa.tuner_floor(mags, gf, offsets=offs)
    f, s, c = aa.tuner_pitches(mags, eff, 48000 / 2048)
torch.cuda.synchronize()
print("mean notes per frame", float(c.float().mean()))

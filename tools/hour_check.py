import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import audio_analyzer_rs_amd as aa, oracle
L = 3600 * 48000
x = aa.signals.synth_corpus(0, L)
xd = torch.from_numpy(x).cuda()
plan = aa.Stft(1024, 256, output="mag")
nf = plan.num_frames(L)
t0 = time.perf_counter()
out = plan(xd)
torch.cuda.synchronize()
print("1 h corpus:", nf, "frames in", round((time.perf_counter() - t0) * 1e3, 2), "ms (incl. alloc)")
for f in (0, 1, 337498, 674990, nf - 1):
    ref = oracle.stft(x[f * 256: f * 256 + 1024], 1024, 256, kind="mag")[0]
    got = out[f].cpu().numpy()
    print(f, float(np.abs(got - ref).max() / ref.max()))
plan2 = aa.Stft(2048, 512, output="logmel", sample_rate=48000, n_mels=128)
o2 = plan2(xd); torch.cuda.synchronize(); print("logmel rows", tuple(o2.shape), bool(torch.isfinite(o2).all()))

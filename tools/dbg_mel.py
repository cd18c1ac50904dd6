import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import audio_analyzer_rs_amd as aa
import oracle
tr = oracle.truth
n, hop, sr, n_mels, fmin, fmax = 1024, 256, 48000, 40, 0.0, 12000.0
x = aa.signals.synth_corpus(9, n + 777 * hop)
xd = torch.from_numpy(x).cuda()
P = np.abs(tr.stft_complex(x, n, hop, oracle.hann_window(n))) ** 2
M = np.maximum(P @ tr.mel_filterbank(sr, n, n_mels, fmin, fmax).T, 1e-10)
def nerr(a, b):
    return np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)
for name, env in (("resident", {}), ("arith", {"AA_MEL_NO_RESIDENT": "1"}), ("gemm", {"AA_MEL_GEMM": "1"})):
    for k in ("AA_MEL_NO_RESIDENT", "AA_MEL_GEMM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    got = aa.Stft(n, hop, output="logmel", sample_rate=sr, n_mels=n_mels, fmin=fmin, fmax=fmax)(xd).cpu().numpy()
    e = nerr(10.0 ** (got.astype(np.float64) / 10.0), M)
    bad = np.nonzero(e > 2e-5)[0]
    print(name, "max", e.max(), "bad frames", bad[:20], len(bad))
    if len(bad):
        f = bad[0]
        d = np.abs(10.0 ** (got[f].astype(np.float64) / 10.0) - M[f]) / M[f].max()
        print("  frame", f, "bad mels", np.nonzero(d > 2e-5)[0][:40], "got", got[f][:6], "ref", 10*np.log10(M[f][:6]))

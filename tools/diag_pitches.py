"""Frames of the pitch test's signal (and of the 10-min corpus head) in which product and restatement disagree: prints
both note lists, so that the kind of decision that flipped can be read off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import audio_analyzer_rs_amd as aa
import oracle
from test_oracle import _chord_signal
sr, n, hop = 48000, 2048, 512
sigs = {"chord+corpus": np.concatenate([_chord_signal(seed=21), aa.signals.synth_corpus(5, 3 * sr)[: 3 * sr]]).astype(np.float32),
        "corpus 60 s": aa.signals.synth_corpus(0, 60 * sr)}
for name, x in sigs.items():
    mags = aa.Stft(n, hop, output="mag")(torch.from_numpy(x).cuda())
    nf, nb = mags.shape
    gf = torch.full((nf,), float(10.0 ** (-60.0 / 20.0) * nb / 2.0), device="cuda")
    eff, _ = aa.tuner_floor(mags, gf)
    f, s, c = [v.cpu().numpy() for v in aa.tuner_pitches(mags, eff, sr / n)]
    rf, rs, rc = oracle.extract_pitches(mags.cpu().numpy(), eff.cpu().numpy(), sr / n)
    bad = np.nonzero(c != rc)[0]
    same = c == rc
    ferr = np.abs(f[same] - rf[same]) / np.maximum(np.abs(rf[same]), 1e-9)
    serr = np.abs(s[same] - rs[same]) / np.maximum(np.abs(rs[same]), 1e-6)
    order = int((np.abs(f[same] - rf[same]) > 1e-3 * np.maximum(rf[same], 1)).any(axis=1).sum())
    print(f"{name}: {nf} frames, counts differ in {len(bad)}, same-count frames with a different note list: {order}, "
          f"max rel err freq {ferr.max():.2e} score {serr.max():.2e}, mean notes {rc.mean():.2f}")
    for t in bad[:12]:
        print("  frame", t, "gpu", [(round(float(a), 3), round(float(b), 5)) for a, b in zip(f[t][:c[t]], s[t][:c[t]])])
        print("        ", "cpu", [(round(float(a), 3), round(float(b), 5)) for a, b in zip(rf[t][:rc[t]], rs[t][:rc[t]])])
    # same notes, different score: which harmonic window sits on an integer boundary?
    rel = np.abs(s - rs) / np.maximum(np.abs(rs), 1e-6)
    tt, jj = np.nonzero((rel > 1e-4) & same[:, None])
    for t, j in list(zip(tt, jj))[:8]:
        fb_g, fb_c = np.float32(f[t, j]) / np.float32(sr / n), np.float32(rf[t, j]) / np.float32(sr / n)
        near = [(k, float(np.float32(fb_c) * np.float32(k))) for k in range(2, 15)]
        near = [(k, v, abs(v - round(v))) for k, v in near]
        k, v, d = min(near, key=lambda z: z[2])
        print(f"  frame {t} note {j}: freq gpu {f[t, j]!r} cpu {rf[t, j]!r} score gpu {s[t, j]!r} cpu {rs[t, j]!r}; "
              f"closest harmonic to an integer bin: n = {k}, frac * n = {v!r} (distance {d:.2e})")

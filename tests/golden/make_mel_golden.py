"""Generates tests/golden/mel_golden.npz — inputs and expected log-mel / MFCC rows of the BUILD-DEFINED
post-ops (BASELINE configs[2] and configs[4]; the reference has no mel stage: parity unpinned by it).
The numbers come from the float64 definition in oracle/truth.py (Slaney mel scale and normalisation,
10 log10(max(., 1e-10)), orthonormal DCT-II); committing them pins that definition, the C-side tables
and the HIP path against a reviewable set of values.

Run from the repo root:  python tests/golden/make_mel_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import truth  # noqa: E402
from audio_analyzer_rs_amd import signals  # noqa: E402

SR, N_MELS, N_MFCC, N_FRAMES = 48000, 128, 40, 20


def cases():
    rng = np.random.default_rng(2024)
    for n, hop in ((2048, 512), (1024, 256)):
        L = n + (N_FRAMES - 1) * hop
        dither = (1e-3 * rng.standard_normal(L)).astype(np.float32)
        yield n, hop, "corpus", signals.synth_corpus(4321, L)
        yield n, hop, "sine440", (signals.sine(440.0, L) + dither).astype(np.float32)
        yield n, hop, "chirp", (signals.chirp(50.0, 12000.0, L) + dither).astype(np.float32)


def main():
    out = {}
    dct = truth.dct2_ortho(N_MFCC, N_MELS)
    for n, hop, name, x in cases():
        P = np.abs(truth.stft_complex(x, n, hop, oracle.hann_window(n))) ** 2
        mel = P @ truth.mel_filterbank(SR, n, N_MELS).T                       # float64 [frames, 128]
        logmel = 10.0 * np.log10(np.maximum(mel, 1e-10))
        key = f"n{n}_{name}"
        out[key + "_x"] = x
        out[key + "_mel"] = mel
        out[key + "_mfcc"] = logmel @ dct.T
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mel_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()

"""Generates tests/golden/stft_golden.npz — small input/expected-output vectors.

The reference holds NO fixtures for this path and cannot be run
here (Rust, no toolchain), so these vectors come from the build's own float64 truth
(oracle/truth.py: f32 window from the C restatement + f32 window multiply as in stft.rs:641-648, 296-299, then
numpy.fft.rfft in float64).  They pin the C oracle and the HIP path against a committed,
reviewable set of numbers; parity with the Rust binary itself stays "unpinned".

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import truth  # noqa: E402
from audio_analyzer_rs_amd import signals  # noqa: E402

GEOMETRIES = [(256, 64), (1024, 256), (2048, 512)]   # onset.rs:122-123, BASELINE, stft.rs:169-170
N_FRAMES = 2


def make_signals(n, hop):
    L = n + (N_FRAMES - 1) * hop
    t = np.arange(L, dtype=np.float64)
    sig = {}
    imp = np.zeros(L, np.float32)
    imp[5] = 1.0
    sig["impulse"] = imp
    sig["dc"] = np.full(L, 0.5, np.float32)
    sig["nyquist"] = (0.5 * np.cos(np.pi * t)).astype(np.float32)
    kb = n // 8
    sig["sine_on_bin"] = (0.8 * np.sin(2 * np.pi * kb * t / n)).astype(np.float32)
    sig["sine_off_bin"] = (0.8 * np.sin(2 * np.pi * (kb + 0.37) * t / n + 0.3)).astype(np.float32)
    sig["lcg_noise"] = signals.lcg_noise(12345, 0, L).astype(np.float32)
    sig["silence"] = np.zeros(L, np.float32)
    sig["corpus"] = signals.synth_corpus(1000, L)
    return sig


def main():
    out = {}
    for n, hop in GEOMETRIES:
        # window from the C restatement (glibc cosf, the call Rust's f32::cos lowers to);
        # stored so the product's aa_hann_window can be pinned bit-for-bit.
        w = oracle.hann_window(n)
        out[f"n{n}_window"] = w
        for name, x in make_signals(n, hop).items():
            key = f"n{n}_{name}"
            out[key + "_x"] = x
            out[key + "_X"] = truth.stft_complex(x, n, hop, w)   # complex128 [frames, n/2+1]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stft_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()

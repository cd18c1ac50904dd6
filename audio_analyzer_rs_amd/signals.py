"""Synthetic 48 kHz mono f32 test/bench signals (host side, numpy).

The corpus follows BASELINE.md / SURVEY.md §8d: a sum of 8 sines (amplitudes 0.05-0.2,
frequencies log-spaced 55 Hz - 7 kHz) plus uniform noise of +-0.01 drawn from the
reference's own LCG (src/generators/metronome.rs:56-57), scaled so the peak cannot
exceed 0.97 (the reference's headroom clamp, src/audio_io/dynamics.rs:319-330).

Every sample is a pure function of its absolute index, so a rank can synthesise just
its own shard (`synth_corpus(start, count)`) and the shards agree bit-for-bit with a
single-process run — no broadcast of the input is needed for the multi-GPU bench.
"""
import numpy as np

LCG_A = 1103515245
LCG_C = 12345
LCG_MASK = 0x7FFFFFFF

N_SINES = 8
SINE_FREQS = np.geomspace(55.0, 7000.0, N_SINES)
SINE_AMPS = np.linspace(0.2, 0.05, N_SINES)
NOISE_AMP = 0.01
# deterministic bound on the peak (all sines in phase + full noise) -> <= 0.97
CORPUS_SCALE = 0.97 / (float(SINE_AMPS.sum()) + NOISE_AMP)


def _lcg_compose(a1, c1, a2, c2):
    """(x -> a2*(a1*x + c1) + c2) mod 2^31 as a single affine map."""
    return (a1 * a2) & LCG_MASK, (a2 * c1 + c2) & LCG_MASK


def lcg_affine_pow(k):
    """Affine map (A, C) of k LCG steps: s_k = (A*s_0 + C) & mask."""
    a, c = 1, 0
    ba, bc = LCG_A & LCG_MASK, LCG_C
    while k:
        if k & 1:
            a, c = _lcg_compose(a, c, ba, bc)
        ba, bc = _lcg_compose(ba, bc, ba, bc)
        k >>= 1
    return a, c


def lcg_states(seed, start, count):
    """States s_{start+1} .. s_{start+count} of the reference LCG started at `seed`
    (s_{k+1} = (s_k * 1103515245 + 12345) & 0x7FFFFFFF), vectorised by doubling."""
    if count <= 0:
        return np.empty(0, dtype=np.uint64)
    a0, c0 = lcg_affine_pow(start)
    s = np.uint64(((a0 * (seed & LCG_MASK)) + c0) & LCG_MASK)   # s_start
    A = np.empty(count, dtype=np.uint64)
    C = np.empty(count, dtype=np.uint64)
    A[0], C[0] = LCG_A & LCG_MASK, LCG_C                        # one step
    n = 1
    mask = np.uint64(LCG_MASK)
    while n < count:
        m = min(n, count - n)
        an, cn = int(A[n - 1]), int(C[n - 1])                  # map of n steps
        # steps n+1 .. n+m = (first 1..m steps) after n steps
        A[n:n + m] = (A[:m] * np.uint64(an)) & mask
        C[n:n + m] = (A[:m] * np.uint64(cn) + C[:m]) & mask
        n += m
    return (A * s + C) & mask


def lcg_noise(seed, start, count):
    """Zero-mean uniform noise in [-1, 1) from the LCG states (float64)."""
    return lcg_states(seed, start, count).astype(np.float64) / float(1 << 30) - 1.0


def synth_corpus(start, count, sample_rate=48000, seed=12345, chunk=1 << 22):
    """Samples [start, start+count) of the synthetic corpus, float32."""
    out = np.empty(count, dtype=np.float32)
    two_pi = 2.0 * np.pi
    for c0 in range(0, count, chunk):
        n = min(chunk, count - c0)
        idx = np.arange(start + c0, start + c0 + n, dtype=np.float64)
        acc = NOISE_AMP * lcg_noise(seed, start + c0, n)
        for f, a in zip(SINE_FREQS, SINE_AMPS):
            acc += a * np.sin(two_pi * f / sample_rate * idx)
        out[c0:c0 + n] = (CORPUS_SCALE * acc).astype(np.float32)
    return out


def sine(freq, count, sample_rate=48000, amp=0.5, start=0):
    idx = np.arange(start, start + count, dtype=np.float64)
    return (amp * np.sin(2.0 * np.pi * freq / sample_rate * idx)).astype(np.float32)


def chirp(f0, f1, count, sample_rate=48000, amp=0.5):
    """Linear chirp f0 -> f1 over `count` samples (the reference has no sweep generator;
    this one is build-written for BASELINE config 5)."""
    t = np.arange(count, dtype=np.float64) / sample_rate
    dur = count / sample_rate
    phase = 2.0 * np.pi * (f0 * t + 0.5 * (f1 - f0) / dur * t * t)
    return (amp * np.sin(phase)).astype(np.float32)


def pcm16_roundtrip(x):
    """What `rec start` writes and a WAV reader returns (src/audio_io/recorder.rs:81-85):
    (clamp(s,-1,1) * 32767) as i16, decoded as i16 / 32768."""
    q = (np.clip(x, -1.0, 1.0) * np.float32(32767.0)).astype(np.int16)   # `as i16` truncates
    return (q.astype(np.float32) / np.float32(32768.0)).astype(np.float32)

"""float64 'truth' for the STFT path (numpy) — TEST INFRASTRUCTURE ONLY.

The window and the window multiply are formed with the reference's f32
semantics (src/audio_io/stft.rs:641-648, 296-299); only the transform and the
magnitude are evaluated in float64.  Any correct f32 FFT (rustfft included)
sits ~1e-7 max-normalised from this, so agreement with it to 5e-6 implies
agreement with the Rust CPU path to 1e-5.

Also holds the build-defined post-op definitions (log-power, mel, MFCC) that
the reference does NOT contain — parity for those is unpinned by the reference.
"""
import numpy as np


def hann_f32(n):
    """Periodic Hann with every step rounded to f32 (numpy float32 arithmetic).

    np.cos on float32 may differ from libm cosf by 1 ulp on rare arguments;
    tests compare the C oracle's table to this with a small ulp allowance.
    """
    i = np.arange(n, dtype=np.float32)
    x = i / np.float32(n)
    arg = np.float32(2.0) * np.float32(np.pi) * x
    return (np.float32(0.5) - np.float32(0.5) * np.cos(arg, dtype=np.float32)).astype(np.float32)


def num_frames(n_samples, n_fft, hop):
    return 0 if n_samples < n_fft else (n_samples - n_fft) // hop + 1


def frames_f32(x, n_fft, hop, window):
    """Windowed frames with the single f32 rounding of stft.rs:296-299."""
    x = np.asarray(x, dtype=np.float32)
    nf = num_frames(x.shape[0], n_fft, hop)
    idx = np.arange(nf)[:, None] * hop + np.arange(n_fft)[None, :]
    return (x[idx] * np.asarray(window, dtype=np.float32)[None, :]).astype(np.float32)


def stft_complex(x, n_fft, hop, window):
    return np.fft.rfft(frames_f32(x, n_fft, hop, window).astype(np.float64), axis=-1)


def stft_mag(x, n_fft, hop, window):
    return np.abs(stft_complex(x, n_fft, hop, window))


def max_normalised_error(out, ref):
    """Per frame: max_k |out-ref| / max_k |ref| (SURVEY.md §8c tolerance definition)."""
    out = np.asarray(out)
    ref = np.asarray(ref)
    num = np.max(np.abs(out.astype(ref.dtype) - ref), axis=-1)
    den = np.max(np.abs(ref), axis=-1)
    den = np.where(den == 0, 1.0, den)
    return num / den


# ---- build-defined post-ops (absent from the reference; parity unpinned) ----

def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz,
                    min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep,
                    f / f_sp)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """Slaney-scale, Slaney-normalised triangular filters, float64 [n_mels, n_fft/2+1]."""
    if fmax is None:
        fmax = sr / 2.0
    nb = n_fft // 2 + 1
    fft_freqs = np.arange(nb, dtype=np.float64) * sr / n_fft
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    w = np.zeros((n_mels, nb), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    return w * enorm[:, None]


def dct2_ortho(n_out, n_in):
    """DCT-II orthonormal basis, float64 [n_out, n_in]."""
    k = np.arange(n_out)[:, None]
    n = np.arange(n_in)[None, :]
    b = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in)) * np.sqrt(2.0 / n_in)
    b[0] *= np.sqrt(0.5)
    return b


LOG_FLOOR = 1e-10


def log_power(power):
    return 10.0 * np.log10(np.maximum(power, LOG_FLOOR))

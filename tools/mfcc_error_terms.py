"""configs[4] (40-coefficient MFCC, 1024 / 256, 128 mels): where the distance to the float64 definition comes from.
Runs on the GPU box.  For the 440 Hz sine and the 50 Hz -> 12 kHz chirp (+ the -60 dB dither of the test):
  e_total   product MFCC rows against the float64 definition (the bench's err_sine / err_chirp)
  e_dct     product MFCC rows against the float64 DCT of the PRODUCT's own log-mel rows: the DCT stage alone
  e_logmel  float64 DCT of (product log-mel rows - float64 log-mel rows): everything in front of the DCT
  e_mel     the same with the product's log-mel rows replaced by float64 log10 of the product's POWER rows contracted in
            float64 with the f32 filterbank: FFT + power alone (mel sums and logs exact)
All max-normalised per frame like the bench figure (max over frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audio_analyzer_rs_amd as aa

def truth_frames(x, n, hop):
    i = np.arange(n, dtype=np.float32)
    w = (np.float32(0.5) - np.float32(0.5) * np.cos(np.float32(2.0) * np.float32(np.pi) * (i / np.float32(n)), dtype=np.float32)).astype(np.float32)
    nf = (len(x) - n) // hop + 1
    idx = np.arange(nf)[:, None] * hop + np.arange(n)[None, :]
    return (x[idx] * w[None, :]).astype(np.float32)

def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64); f_sp = 200.0 / 3; mlh = 1000.0; mlm = mlh / f_sp; ls = np.log(6.4) / 27.0
    return np.where(f >= mlh, mlm + np.log(np.maximum(f, 1e-30) / mlh) / ls, f / f_sp)
def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64); f_sp = 200.0 / 3; mlh = 1000.0; mlm = mlh / f_sp; ls = np.log(6.4) / 27.0
    return np.where(m >= mlm, mlh * np.exp(ls * (m - mlm)), f_sp * m)
def filterbank(sr, n_fft, n_mels):
    nb = n_fft // 2 + 1
    ff = np.arange(nb, dtype=np.float64) * sr / n_fft
    hz = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fd = np.diff(hz); ramps = hz[:, None] - ff[None, :]
    w = np.zeros((n_mels, nb))
    for i in range(n_mels):
        w[i] = np.maximum(0.0, np.minimum(-ramps[i] / fd[i], ramps[i + 2] / fd[i + 1]))
    return w * (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[:, None]
def dct2(n_out, n_in):
    k = np.arange(n_out)[:, None]; n = np.arange(n_in)[None, :]
    b = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in)) * np.sqrt(2.0 / n_in); b[0] *= np.sqrt(0.5)
    return b
def nerr(a, b):
    num = np.max(np.abs(a - b), axis=-1); den = np.max(np.abs(b), axis=-1)
    return (num / np.where(den == 0, 1.0, den))

sr, n, hop, n_mels, n_mfcc = 48000, 1024, 256, 128, 40
fb, D = filterbank(sr, n, n_mels), dct2(n_mfcc, n_mels)
rng = np.random.default_rng(1)
dither = (1e-3 * rng.standard_normal(10 * sr)).astype(np.float32)
for name, sig in (("sine", aa.signals.sine(440.0, 10 * sr)), ("chirp", aa.signals.chirp(50.0, 12000.0, 10 * sr))):
    x = (sig + dither).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    mfcc = aa.Stft(n, hop, output="mfcc", sample_rate=sr, n_mels=n_mels, n_mfcc=n_mfcc)(xd).cpu().numpy().astype(np.float64)
    lm = aa.Stft(n, hop, output="logmel", sample_rate=sr, n_mels=n_mels)(xd).cpu().numpy().astype(np.float64)
    pw = aa.Stft(n, hop, output="power")(xd).cpu().numpy().astype(np.float64)
    P = np.abs(np.fft.rfft(truth_frames(x, n, hop).astype(np.float64), axis=-1)) ** 2
    L = 10.0 * np.log10(np.maximum(P @ fb.T, 1e-10))
    ref = L @ D.T
    den = np.max(np.abs(ref), axis=-1)
    e_total = nerr(mfcc, ref)
    e_dct = np.max(np.abs(mfcc - lm @ D.T), axis=-1) / den
    e_logmel = np.max(np.abs((lm - L) @ D.T), axis=-1) / den
    Lp = 10.0 * np.log10(np.maximum(pw @ fb.astype(np.float32).astype(np.float64).T, 1e-10))
    e_mel = np.max(np.abs((Lp - L) @ D.T), axis=-1) / den
    e_gather = np.max(np.abs((lm - Lp) @ D.T), axis=-1) / den
    t = int(np.argmax(e_total))
    print(f"{name:6s} e_total {e_total.max():.3e} (frame {t})  e_dct {e_dct.max():.3e}  e_logmel {e_logmel.max():.3e}  "
          f"e_fft_power {e_mel.max():.3e}  e_melsum_log {e_gather.max():.3e}")
    dl = lm[t] - L[t]
    worst = np.argsort(-np.abs(dl))[:6]
    print("   worst frame: log-mel deviations [dB] at mels", worst.tolist(), np.round(dl[worst], 6).tolist(), " L there", np.round(L[t][worst], 2).tolist(),
          " max|mfcc|", round(float(den[t]), 2))
    dlp = Lp[t] - L[t]
    print("   same mels, from the power rows alone:", np.round(dlp[worst], 6).tolist())

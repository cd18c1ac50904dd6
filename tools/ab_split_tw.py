"""r5 experiment: the real-FFT split twiddle -i W_1024^k as a two-float number (hi + lo) for the power-type
rows.  Per library (shipped / -DAA_SPLIT_TW2=1), in a process of its own: the bench's err_sine / err_chirp, and over the 21
chirp cases of tests/test_gpu_parity.py::test_mfcc_error_terms_and_chirp_sweep the FFT's share e_fft (float64 post-ops of the
product's POWER rows against the float64 definition) and the product's MFCC error e_all; the launch times of log-mel and MFCC.
Usage: python tools/ab_split_tw.py lib_a.so lib_b.so"""
import os, subprocess, sys
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import audio_analyzer_rs_amd as aa
aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
from oracle import truth as tr
import oracle
sr, n, hop, n_mels, n_mfcc = 48000, 1024, 256, 128, 40
fb, dct = tr.mel_filterbank(sr, n, n_mels), tr.dct2_ortho(n_mfcc, n_mels)
win = oracle.hann_window(n)
post = lambda P: (10.0 * np.log10(np.maximum(P @ fb.T, 1e-10))) @ dct.T
nerr = tr.max_normalised_error
plan = aa.Stft(n, hop, output="mfcc", sample_rate=sr, n_mels=n_mels, n_mfcc=n_mfcc)
pplan = aa.Stft(n, hop, output="power")
rng = np.random.default_rng(1)
dither = (1e-3 * rng.standard_normal(10 * sr)).astype(np.float32)
for name, sig in (("sine", aa.signals.sine(440.0, 10 * sr)), ("chirp", aa.signals.chirp(50.0, 12000.0, 10 * sr))):
    x = (sig + dither).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    ref = post(np.abs(tr.stft_complex(x, n, hop, win)) ** 2)
    got = plan(xd).cpu().numpy().astype(np.float64)
    pw = pplan(xd).cpu().numpy().astype(np.float64)
    print(f"  {name}: err {nerr(got, ref).max():.3e}  (FFT share {nerr(post(pw), ref).max():.3e})")
rows = []
for f1 in (3000.0, 6000.0, 8000.0, 12000.0, 16000.0, 20000.0, 23000.0):
    for seed in (1, 2, 3):
        r2 = np.random.default_rng(seed)
        x = (aa.signals.chirp(50.0, f1, 4 * sr) + (1e-3 * r2.standard_normal(4 * sr)).astype(np.float32)).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        ref = post(np.abs(tr.stft_complex(x, n, hop, win)) ** 2)
        got = plan(xd).cpu().numpy().astype(np.float64)
        pw = pplan(xd).cpu().numpy().astype(np.float64)
        e_all, e_fft = nerr(got, ref), nerr(post(pw), ref)
        rows.append((e_all.max(), np.quantile(e_all, 0.99), e_fft.max(), np.quantile(e_fft, 0.99)))
w = np.array(rows)
print(f"  sweep of 21: e_all max {w[:,0].max():.3e} median-of-max {np.median(w[:,0]):.3e} worst p99 {w[:,1].max():.3e} | "
      f"e_fft max {w[:,2].max():.3e} median-of-max {np.median(w[:,2]):.3e} worst p99 {w[:,3].max():.3e}")
x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
st = torch.cuda.current_stream()
for kind, kw in (("logmel", dict(n_mels=128)), ("mfcc", dict(n_mels=128, n_mfcc=40)), ("power", {})):
    p = aa.Stft(n, hop, output=kind, **kw)
    nf = p.num_frames(len(x)); out = torch.empty((nf, p.cols), device="cuda")
    for _ in range(300): p.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(1000): p.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    print(f"  {kind}: {e0.elapsed_time(e1):.2f} us per launch")
'''
for rnd in range(2):
    for lib in sys.argv[1:]:
        print(f"round {rnd} {os.path.basename(lib)}", flush=True)
        print(subprocess.check_output([sys.executable, "-c", CHILD, lib], stderr=subprocess.DEVNULL).decode(), flush=True)

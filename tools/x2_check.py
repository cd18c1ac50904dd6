"""Diagnostic (GPU box): parity of stft_wave1024x2_kernel against the float64 truth over ragged frame counts and
sub-ranges, then launch times of the two 1024 / 256 magnitude kernels (AA_W1024_X2=0 | 1 at plan creation) in
one process, interleaved.  numpy truth only - tools never import oracle/.
Usage: python tools/x2_check.py [seconds_per_leg]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_analyzer_rs_amd as aa


def hann_f32(n):
    i = np.arange(n, dtype=np.float32)
    return (np.float32(0.5) - np.float32(0.5) * np.cos(np.float32(2.0) * np.float32(np.pi) * (i / np.float32(n)),
                                                       dtype=np.float32)).astype(np.float32)


def truth_mag(x, n, hop):
    nf = (len(x) - n) // hop + 1
    idx = np.arange(nf)[:, None] * hop + np.arange(n)[None, :]
    return np.abs(np.fft.rfft((x[idx] * hann_f32(n)[None, :]).astype(np.float32).astype(np.float64), axis=-1))


def nerr(got, ref):
    den = np.max(np.abs(ref), axis=-1)
    return np.max(np.abs(got.astype(ref.dtype) - ref), axis=-1) / np.where(den == 0, 1.0, den)


os.environ["AA_W1024_X2"] = "1"
plan = aa.Stft(1024, 256, output="mag")
print("kernel:", plan.kernel, flush=True)
worst = 0.0
for nf in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 23, 24, 25, 33, 100, 1000, 4099, 20011):
    x = aa.signals.synth_corpus(nf, 1024 + (nf - 1) * 256 + 13)
    buf = torch.full((nf * 513 + 64,), float("nan"), device="cuda")
    out = buf[:nf * 513].view(nf, 513)
    got = plan.execute_device(torch.from_numpy(x).cuda(), 0, nf, out=out)
    torch.cuda.synchronize()
    assert torch.isnan(buf[nf * 513:]).all(), "wrote past the slab"
    g = got.cpu().numpy()
    assert np.isfinite(g).all(), nf
    e = float(nerr(g, truth_mag(x, 1024, 256)).max())
    worst = max(worst, e)
    print(f"  nf {nf:6d}  err {e:.2e}", flush=True)
    assert e < 5e-6, (nf, e)
# sub-ranges
x = aa.signals.synth_corpus(3, 1024 + 999 * 256)
xd = torch.from_numpy(x).cuda()
T = truth_mag(x, 1024, 256)
for f0, f1 in ((0, 1000), (1, 1000), (7, 13), (8, 9), (123, 999), (500, 501)):
    g = plan.execute_device(xd, f0, f1).cpu().numpy()
    e = float(nerr(g, T[f0:f1]).max())
    worst = max(worst, e)
    assert e < 5e-6, (f0, f1, e)
print(f"parity ok, worst {worst:.2e}", flush=True)

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
NS = 28_800_000
xd = torch.from_numpy(aa.signals.synth_corpus(0, NS)).cuda()
plans = {}
for v in ("0", "1"):
    os.environ["AA_W1024_X2"] = v
    plans[v] = aa.Stft(1024, 256, output="mag")
    print("x2 =", v, plans[v].kernel)
nf = plans["1"].num_frames(NS)
out = torch.empty((nf, 513), dtype=torch.float32, device="cuda")
a = plans["0"].execute_device(xd, 0, nf, out=out).clone()
b = plans["1"].execute_device(xd, 0, nf, out=out)
torch.cuda.synchronize()
print("full-size max |a-b| / max:", float((a - b).abs().max() / a.abs().max()))
del a
st = torch.cuda.current_stream()
for rnd in range(3):
    for v in ("0", "1"):
        p = plans[v]
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            for i in range(50):
                p.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
            torch.cuda.synchronize()
        n_l = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            for i in range(200):
                p.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
            n_l += 200
            torch.cuda.synchronize()
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / n_l
        print(f"round {rnd} x2={v}: {us:.2f} us/launch  frac {nf * 3076 / us / 8e6:.4f}", flush=True)

"""Parity of library builds (variants of the kernels under A/B): every row of a few ragged sizes
against the float64 truth, each build in its own process.
Usage: python tools/variant_check.py libA.so [libB.so ...]     (diagnostic, not shipped)"""
import os
import subprocess
import sys

CHILD = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import audio_analyzer_rs_amd as aa
aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
worst = {}


def hann_f32(n):          # periodic Hann, every step in f32 (stft.rs:641-648)
    i = np.arange(n, dtype=np.float32)
    return (np.float32(0.5) - np.float32(0.5) * np.cos(np.float32(2.0) * np.float32(np.pi) * (i / np.float32(n)), dtype=np.float32)).astype(np.float32)


def spectrum64(x, n, hop):  # float64 rfft of the f32-windowed frames (numpy only: tools never touch oracle/)
    nf = (len(x) - n) // hop + 1
    idx = np.arange(nf)[:, None] * hop + np.arange(n)[None, :]
    return np.fft.rfft((x[idx] * hann_f32(n)[None, :]).astype(np.float32).astype(np.float64), axis=-1)


def nerr(got, ref):
    den = np.max(np.abs(ref), axis=-1)
    return np.max(np.abs(got.astype(ref.dtype) - ref), axis=-1) / np.where(den == 0, 1.0, den)

geoms = [tuple(int(v) for v in g.split("/")) for g in os.environ.get("VC_GEOMS", "1024/256").split(",")]
for n, hop in geoms:
  for kind in ("mag", "complex"):
    plan = aa.Stft(n, hop, output=kind)
    for nf in (1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 33, 1000, 4099, 20011):
        x = aa.signals.synth_corpus(nf, n + (nf - 1) * hop)
        buf = torch.full((nf * plan.cols + 64,), float("nan"), device="cuda")
        out = buf[:nf * plan.cols].view(nf, plan.cols)
        got = plan.execute_device(torch.from_numpy(x).cuda(), 0, nf, out=out)
        torch.cuda.synchronize()
        assert torch.isnan(buf[nf * plan.cols:]).all(), "wrote past the slab"
        got = got.cpu().numpy()
        ref = spectrum64(x, n, hop)
        if kind == "mag":
            ref = np.abs(ref)
        e = float(nerr(got, ref).max())
        assert np.isfinite(got.view(np.float32)).all(), (n, kind, nf)
        worst[(n, kind)] = max(worst.get((n, kind), 0.0), e)
print(" ".join(f"{n}/{k}:{e:.2e}" for (n, k), e in worst.items()), "OK" if max(worst.values()) < 5e-6 else "FAIL")
"""

for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or ["<no output>"])[-1]
    print(f"{os.path.basename(lib):28s} rc={r.returncode} {tail}", flush=True)
    if r.returncode:
        print(r.stderr[-1500:], flush=True)

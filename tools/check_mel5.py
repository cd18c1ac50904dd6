"""r5 experiment: the fifth-wave log-mel kernel (-DAA_MEL5=1) against the shipped one: same rows bit for bit over ragged
frame counts (each library in a process of its own, rows hashed), then the launch times.  Usage: python tools/check_mel5.py a.so b.so"""
import os, subprocess, sys
CHILD = r'''
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, os.getcwd())
import audio_analyzer_rs_amd as aa
aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
n, hop = 1024, 256
plan = aa.Stft(n, hop, output="logmel", sample_rate=48000, n_mels=128)
h = hashlib.sha256()
for nf in (1, 2, 7, 8, 9, 15, 16, 17, 31, 33, 100, 1000, 4099, 20011, 112497):
    x = aa.signals.synth_corpus(nf, n + (nf - 1) * hop)
    buf = torch.full((nf * plan.cols + 64,), float("nan"), device="cuda")
    out = buf[:nf * plan.cols].view(nf, plan.cols)
    plan.execute_device(torch.from_numpy(x).cuda(), 0, nf, out=out)
    torch.cuda.synchronize()
    assert torch.isnan(buf[nf * plan.cols:]).all(), "wrote past the slab"
    assert torch.isfinite(out).all(), ("non-finite rows", nf)
    h.update(out.cpu().numpy().tobytes())
print("rows", h.hexdigest()[:16], plan.mel_path)
'''
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True, timeout=240)
    print(os.path.basename(lib), "rc", r.returncode, (r.stdout.strip().splitlines() or ["<none>"])[-1], flush=True)
    if r.returncode:
        print(r.stderr[-1200:], flush=True)

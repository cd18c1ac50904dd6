"""Bit-for-bit comparison of the shipped library with a build whose counted s_waitcnt are all full waits
(-DAA_SAFE_WAITS=1, tools/bin/libaa_safe.so): every wave kernel and output kind, full-size launches
back to back (stores of launch i in flight under launch i + 1), several repetitions.  A counted wait
that let an LDS-DMA piece, a streamed weight or a hand-issued LDS load slip would change bits.
Usage (GPU box): python tools/check_counted_waits.py audio_analyzer_rs_amd/libaa_stft.so tools/bin/libaa_safe.so
Each library runs in its own process (the loader binds one library per process)."""
import hashlib
import os
import subprocess
import sys

CHILD = r"""
import hashlib, os, sys, torch
sys.path.insert(0, os.getcwd())
import audio_analyzer_rs_amd as aa
aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
cases = [(1024, 256, "mag", {}), (1024, 256, "complex", {}), (1024, 256, "logmel", dict(n_mels=128)),
         (1024, 256, "mfcc", dict(n_mels=128, n_mfcc=40)), (1024, 256, "logpower", {}),
         (2048, 512, "mag", {}), (2048, 512, "complex", {}), (2048, 512, "logmel", dict(n_mels=128)),
         (2048, 512, "logmel", dict(n_mels=40)), (2048, 500, "power", {}),
         (256, 64, "mag", {}), (512, 128, "mag", {}), (128, 32, "complex", {}),
         (1024, 256, "logmel", dict(n_mels=30)), (1024, 256, "mfcc", dict(n_mels=126, n_mfcc=33))]   # the VALU-gather kinds
xs = [torch.from_numpy(aa.signals.synth_corpus(0, 14_400_000 + 4 * 997, seed=12345 + 11 * i)).cuda() for i in range(2)]
for n, hop, kind, kw in cases:
    kw = dict(kw)
    plan = aa.Stft(n, hop, output=kind, **kw)
    kw["kernel"] = plan.kernel + ("/" + plan.mel_path if plan.mel_path else "")
    nf = plan.num_frames(xs[0].shape[0])
    outs = [torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda") for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    h = hashlib.sha256()
    for rep in range(6):
        for i in range(8):
            plan.execute_device(xs[i % 2], 0, nf, out=outs[i % 2], stream=st)
        torch.cuda.synchronize()
        for o in outs:
            h.update(o.cpu().numpy().tobytes())
    print(n, hop, kind, sorted(kw.items()), h.hexdigest()[:24], flush=True)
"""

libs = sys.argv[1:3]
res = []
for lib in libs:
    out = subprocess.check_output([sys.executable, "-c", CHILD, lib]).decode().strip().splitlines()
    res.append(out)
bad = 0
for a, b in zip(*res):
    same = a == b
    bad += not same
    print(("same   " if same else "DIFFER ") + a + ("" if same else "  |  " + b))
print("all identical" if bad == 0 and len(res[0]) == len(res[1]) else f"{bad} case(s) differ")
sys.exit(1 if bad else 0)

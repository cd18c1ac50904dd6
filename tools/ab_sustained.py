"""A/B of library builds under the sustained (power-limited) regime of bench.py: each build runs in
its own process, alternating, one launch (default: the headline 1024/256 magnitudes of 10 min;
AB_NFFT / AB_HOP / AB_OUT select another geometry) back to back.
Usage: python tools/ab_sustained.py libA.so libB.so [libC.so ...] [rounds]   (diagnostic, not shipped)"""
import os
import subprocess
import sys

CHILD = r"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import audio_analyzer_rs_amd as aa
aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
NS = int(os.environ.get("AB_SAMPLES", "28800000"))   # AB_SAMPLES: another corpus length
x = aa.signals.synth_corpus(0, NS)
xd = torch.from_numpy(x).cuda()
ALT = int(os.environ.get("AB_ALT", "1"))          # AB_ALT=2: two input / output pairs alternating (stereo, configs[2])
xs = [xd] + [torch.from_numpy(aa.signals.synth_corpus(0, NS, seed=54321 + i)).cuda() for i in range(ALT - 1)]
n_fft, hop, kind = int(os.environ.get("AB_NFFT", "1024")), int(os.environ.get("AB_HOP", "256")), os.environ.get("AB_OUT", "mag")
extra = dict(n_mels=128) if kind in ("logmel", "mfcc") else {}
if kind == "mfcc":
    extra["n_mfcc"] = 40
plan = aa.Stft(n_fft, hop, output=kind, **extra)
nf = plan.num_frames(xd.shape[0])
outs = [torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda") for _ in range(ALT)]
out = outs[0]
st = torch.cuda.current_stream()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for i in range(50):
        plan.execute_device(xs[i % ALT], 0, nf, out=outs[i % ALT], stream=st.cuda_stream)
    torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(1000):
        plan.execute_device(xs[i % ALT], 0, nf, out=outs[i % ALT], stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1))
print(" ".join("%.2f" % r for r in res))
"""

# a library may carry environment settings for its child process: path@NAME=VALUE[@NAME=VALUE..]
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
for r in range(rounds):
    for lib in libs:
        path, *sets = lib.split("@")
        env = dict(os.environ, **dict(kv.split("=", 1) for kv in sets))
        out = subprocess.check_output([sys.executable, "-c", CHILD, path], env=env, stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
        print(f"round {r} {os.path.basename(lib):32s} us/step: {out}", flush=True)

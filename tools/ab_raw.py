"""A/B of library builds through the bare C ABI (ctypes, four entry points), for builds older than the Python package's
signature table: python tools/ab_raw.py <n_fft> <hop> <output code> libA.so libB.so ... [rounds]
output codes: 0 complex, 1 mag, 2 power, 3 logpower, 4 log-mel (128 bands), 5 MFCC (128 bands, 40 coefficients) (include/aa_stft.h).  One process per library and round, alternating."""
import os, subprocess, sys

CHILD = r"""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from audio_analyzer_rs_amd import signals
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
n_fft, hop, kind = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
class Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_fft", "hop", "window", "output", "sample_rate", "n_mels", "n_mfcc")] + \
               [(n, ctypes.c_float) for n in ("fmin", "fmax", "log_floor")]
lib.aa_stft_plan_create.argtypes = [ctypes.POINTER(Cfg), ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
lib.aa_stft_execute_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
lib.aa_stft_out_cols.argtypes = [ctypes.c_void_p]
x = torch.from_numpy(signals.synth_corpus(0, 28_800_000)).cuda()
cfg = Cfg(n_fft, hop, 0, kind, 48000, 128 if kind >= 4 else 0, 40 if kind == 5 else 0, 0.0, 0.0, 0.0)
plan = ctypes.c_void_p()
assert lib.aa_stft_plan_create(ctypes.byref(cfg), None, ctypes.byref(plan)) == 0
nf = (x.numel() - n_fft) // hop + 1
out = torch.empty((nf, lib.aa_stft_out_cols(plan)), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream()
def go(n):
    for _ in range(n):
        assert lib.aa_stft_execute_device(plan, x.data_ptr(), x.numel(), 0, nf, out.data_ptr(), st.cuda_stream) == 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    go(50); torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); go(1000); e1.record(st); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1))
print(" ".join("%.2f" % r for r in res))
"""
n_fft, hop, kind = sys.argv[1:4]
libs = [a for a in sys.argv[4:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() and len(sys.argv) > 5 else 2
for r in range(rounds):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, lib, n_fft, hop, kind], capture_output=True, text=True)
        print(f"round {r} {os.path.basename(lib):28s} us/step: {(out.stdout.strip().splitlines() or [out.stderr[-200:]])[-1]}", flush=True)

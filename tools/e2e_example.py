"""End-to-end wall time of examples/aa_analyze (the C-ABI-only host program) on a 10-minute recorder
WAV: conditioner -> tuner chain + onset chain -> JSON.  Usage: python tools/e2e_example.py  (diagnostic)"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import audio_analyzer_rs_amd as aa  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "examples", "aa_analyze")
x = aa.signals.synth_corpus(0, 28_800_000)
with tempfile.TemporaryDirectory() as d:
    wav = os.path.join(d, "ten_minutes.wav")
    aa.wav.write_wav(wav, x, 48000)
    for rep in range(3):
        t0 = time.perf_counter()
        out = subprocess.check_output([exe, wav, "120", "2000"], env=dict(os.environ, AA_ANALYZE_TIMING="1") if rep == 2 else None)
        dt = time.perf_counter() - t0
        print(f"run {rep}: {dt:.2f} s wall for 600 s of audio ({600 / dt:.0f} x real time), {len(out)} bytes of JSON")

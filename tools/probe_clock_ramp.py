"""Diagnostic: how long after an idle gap a launch takes to reach its steady time.  One process: sleep, then windows of 50
back-to-back launches (HIP events) for ~1.5 s, for the headline magnitudes and the fused MFCC launch.
Usage (GPU box): python tools/probe_clock_ramp.py [idle_seconds=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

idle = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
st = torch.cuda.current_stream()
for kind, kw in (("mag", {}), ("mfcc", dict(n_mels=128, n_mfcc=40)), ("logmel", dict(n_mels=128))):
    plan = aa.Stft(1024, 256, output=kind, **kw)
    nf = plan.num_frames(xd.shape[0])
    out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize()
        time.sleep(idle)
        t0 = time.perf_counter()
        row = []
        for w in range(24):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(50):
                plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 50 * 1e3))
        print(kind, "after %.0f s idle:" % idle, " ".join("%.0fms:%.1f" % (t * 1e3, us) for t, us in row), flush=True)

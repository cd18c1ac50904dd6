"""Kernel time of the bench workload over a long back-to-back run (per 200-launch window):
shows the cold ramp and whether sustained load settles at a lower clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa
x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
plan = aa.Stft(1024, 256, output="mag")
nf = plan.num_frames(xd.shape[0])
out = torch.empty((nf, 513), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
t_start = time.perf_counter()
evs = []
for w in range(int(os.environ.get("AA_DRIFT_WINDOWS", "60"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        plan.execute_device(xd, 0, nf, out=out, stream=st)
    e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
tot = 0.0
for w, (e0, e1) in enumerate(evs):
    ms = e0.elapsed_time(e1)
    tot += ms
    if w < 10 or w % 5 == 0:
        print(f"window {w:3d}  t={tot:8.1f} ms  kernel {ms / 200 * 1e3:6.1f} us")

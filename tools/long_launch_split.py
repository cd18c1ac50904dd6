"""One long launch against the same frames as several shorter launches (diagnostic): beyond ~0.7-1 GB
touched per launch the wave kernels slow down (address translation reach); python tools/long_launch_split.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

x = torch.from_numpy(aa.signals.synth_corpus(0, 172_800_000)).cuda()
plan = aa.Stft(1024, 256, output="mag")
nf = plan.num_frames(x.shape[0])
out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream()
for parts in (1, 2, 3, 4, 6, 8, 12):
    edges = [nf * i // parts for i in range(parts + 1)]

    def step():
        for a, b in zip(edges[:-1], edges[1:]):
            plan.execute_device(x, a, b, out=out[a:], stream=st.cuda_stream)

    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(200):
        step()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("parts %2d: %.1f us per pass (%d frames), %.1f us per 112497 frames" % (parts, ms * 1e3, nf, ms * 1e3 * 112497 / nf), flush=True)

"""Probe (diagnostic): does the launch-time mode (complex rows ~90 or ~112 us per process) follow the HIP stream /
hardware queue?  Times the same plan and buffers on the default stream and on several fresh streams.
Run on a GPU box: python tools/probe_bimodal_streams.py [kind]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

kind = sys.argv[1] if len(sys.argv) > 1 else "complex"
x = aa.signals.synth_corpus(0, 28_800_000)
plan = aa.Stft(1024, 256, output=kind)
nf = plan.num_frames(len(x))
xd = torch.from_numpy(x).cuda()
out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")


def timeit(st, n=300):
    for _ in range(200):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


d = torch.cuda.current_stream()
t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(d, 50)
print(f"default stream: {timeit(d):.2f} us, {timeit(d):.2f} us")
streams = [torch.cuda.Stream() for _ in range(6)] + [torch.cuda.Stream(priority=-1) for _ in range(2)]
for i, s in enumerate(streams):
    print(f"stream {i} ({s.cuda_stream:#x}{' high priority' if i >= 6 else ''}): {timeit(s):.2f} us, {timeit(s):.2f} us")
print(f"default stream again: {timeit(d):.2f} us")

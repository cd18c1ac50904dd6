"""Probe (diagnostic): several input and output buffers LIVE at the same time in one process (so they
cannot share physical pages); the complex launch timed on every (input, output) combination.  If the time
depends on the buffers, placement matters; if every combination of a process runs at the process's speed,
something else sets the mode.  Run on a GPU box: python tools/probe_bimodal_live.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

x = aa.signals.synth_corpus(0, 28_800_000)
plan = aa.Stft(1024, 256, output="complex")
nf = plan.num_frames(len(x))
st = torch.cuda.current_stream()
ins = [torch.from_numpy(x).cuda() for _ in range(3)]
outs = [torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda") for _ in range(5)]


def timeit(xd, out, n=200):
    for _ in range(100):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(ins[0], outs[0], 50)
for i, xd in enumerate(ins):
    print(f"in {i} @ {xd.data_ptr():#x}: " + "  ".join(f"out{j} {timeit(xd, o):6.2f}" for j, o in enumerate(outs)))
print("out addresses: " + " ".join(f"{o.data_ptr():#x}" for o in outs))


def tm(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# is it the buffer's memory itself?  plain write (fill) and read (sum of a view) bandwidth per output buffer
gb = outs[0].numel() * 4 / 1e9
print("fill  TB/s: " + "  ".join(f"out{j} {gb / tm(lambda o=o: o.fill_(1.0)) * 1e3:5.2f}" for j, o in enumerate(outs)))
print("read  TB/s: " + "  ".join(f"out{j} {gb / tm(lambda o=o: torch.sum(o)) * 1e3:5.2f}" for j, o in enumerate(outs)))
print("copy out0->outj TB/s (r+w): " + "  ".join(f"out{j} {2 * gb / tm(lambda o=o: o.copy_(outs[0])) * 1e3:5.2f}" for j, o in enumerate(outs) if j))

"""Probe (diagnostic): one 3 GB slab, the complex output placed at offsets 0, 64 MB, 128 MB, ... inside it:
which regions of one allocation are "fast" (~90 us) and which "slow" (~112 us) for the 1024/256 complex launch?
Run on a GPU box: python tools/probe_bimodal_slab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

x = aa.signals.synth_corpus(0, 28_800_000)
plan = aa.Stft(1024, 256, output="complex")
nf = plan.num_frames(len(x))
st = torch.cuda.current_stream()
xd = torch.from_numpy(x).cuda()
need = nf * plan.cols
slab = torch.empty(need + (2560 << 20) // 4, dtype=torch.float32, device="cuda")


def timeit(out, n=120):
    for _ in range(60):
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
o0 = slab[:need].view(nf, plan.cols)
while time.time() - t0 < 0.5:
    timeit(o0, 50)
print(f"slab @ {slab.data_ptr():#x}")
line = []
for mb in range(0, 2561, 64):
    o = slab[(mb << 20) // 4: (mb << 20) // 4 + need].view(nf, plan.cols)
    line.append(f"{mb:4d}MB {timeit(o):6.1f}")
    if len(line) == 6:
        print("  ".join(line)); line = []
if line:
    print("  ".join(line))
# finer scan around the first change

"""Probe (diagnostic): libraries (argv) timed on the SAME slow and fast output placements of one process:
a 3 GB slab, output at offset 0 (slow region on the boxes seen) and at the end (fast region).
Usage: python tools/probe_bimodal_libs.py kind lib.so          (one library per process)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa
kind = sys.argv[1]
aa._lib.LIB_PATH = os.path.abspath(sys.argv[2])
x = aa.signals.synth_corpus(0, 28_800_000)
plan = aa.Stft(1024, 256, output=kind)
nf = plan.num_frames(len(x))
st = torch.cuda.current_stream()
xd = torch.from_numpy(x).cuda()
need = nf * plan.cols
slab = torch.empty(need + (2560 << 20) // 4, dtype=torch.float32, device="cuda")
def timeit(out, n=300):
    for _ in range(150):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
o_a = slab[:need].view(nf, plan.cols)
o_b = slab[(2560 << 20) // 4:(2560 << 20) // 4 + need].view(nf, plan.cols)
t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(o_a, 50)
print(f"{os.path.basename(sys.argv[2]):20s} {kind:8s} slab start {timeit(o_a):6.2f} us   slab end {timeit(o_b):6.2f} us   start again {timeit(o_a):6.2f}")

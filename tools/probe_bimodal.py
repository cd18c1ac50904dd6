"""Probe (diagnostic): the 1024/256 complex-output launch runs at ~90 us or ~112 us from one process to the
next on the same box.  Does the mode follow the buffers (addresses / physical pages) or the process?
Re-allocates input and output several times inside ONE process and times 300 launches for each placement;
then keeps one slab and slides the output inside it.  Run on a GPU box: python tools/probe_bimodal.py [kind]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

kind = sys.argv[1] if len(sys.argv) > 1 else "complex"
x = aa.signals.synth_corpus(0, 28_800_000)
plan = aa.Stft(1024, 256, output=kind)
nf = plan.num_frames(len(x))
st = torch.cuda.current_stream()


def timeit(xd, out, n=300):
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


# warm the clocks
xd = torch.from_numpy(x).cuda()
out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")
t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(xd, out, 50)
print(f"first placement: x @ {xd.data_ptr():#x} out @ {out.data_ptr():#x}: {timeit(xd, out):.2f} us")
keep = []
for rep in range(6):
    # new physical pages: drop the cached blocks, hold a filler of varying size so that the next blocks land elsewhere
    del xd, out
    torch.cuda.empty_cache()
    keep.append(torch.empty((rep + 1) * 37 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda"))
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")
    print(f"re-allocation {rep}: x @ {xd.data_ptr():#x} out @ {out.data_ptr():#x}: {timeit(xd, out):.2f} us  (again {timeit(xd, out):.2f})")
# one slab, sliding output
slab = torch.empty(nf * plan.cols + (64 << 20) // 4, dtype=torch.float32, device="cuda")
for off_bytes in (0, 16, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 8 << 20, 32 << 20):
    o = slab[off_bytes // 4: off_bytes // 4 + nf * plan.cols].view(nf, plan.cols)
    print(f"slab offset {off_bytes:>9d}: out @ {o.data_ptr():#x}: {timeit(xd, o):.2f} us")
# same buffers, timed repeatedly with idle gaps
for gap in (0.0, 0.5, 2.0):
    time.sleep(gap)
    print(f"after {gap:.1f} s idle: {timeit(xd, out):.2f} us, then {timeit(xd, out):.2f} us")

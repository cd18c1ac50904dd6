"""Does the magnitude launch care where its INPUT lies?  Copies of the 115-MB sample buffer are allocated one after the
other (all kept) and the headline launch is timed from each into one fixed output.  Argument: number of copies (default 200)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = aa.signals.synth_corpus(0, 28_800_000)
plan = aa.Stft(1024, 256, output="mag")
nf = plan.num_frames(len(x))
out = torch.empty((nf, plan.cols), device="cuda")
st = torch.cuda.current_stream()
src = torch.from_numpy(x).cuda()


def timeit(xd, k=30, warm=8):
    for _ in range(warm):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(k):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


timeit(src, 3000, 300)                                    # the sustained regime first
copies = [src.clone() for _ in range(n)]
ts = [timeit(c) for c in copies]
again = [timeit(c) for c in copies]
s = sorted(ts)
print(f"{n} input copies: min {s[0]:.2f} p10 {s[n // 10]:.2f} median {s[n // 2]:.2f} p90 {s[9 * n // 10]:.2f} max {s[-1]:.2f} us")
best = min(range(n), key=lambda i: ts[i]); worst = max(range(n), key=lambda i: ts[i])
print(f"fastest copy {best}: {ts[best]:.2f} (again {again[best]:.2f}); slowest {worst}: {ts[worst]:.2f} (again {again[worst]:.2f})")
print("correlation first/second pass:", float(torch.corrcoef(torch.tensor([ts, again]))[0, 1]))

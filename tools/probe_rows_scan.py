"""Scan of the device's memory for FAST placements of the complex-rows launch (~90 us into some output
buffers, ~118 us into others): allocates slab after slab of the launch's output size (461 MB) up to a byte budget, times the
launch into each (all stay allocated, so every slab is a different piece of physical memory), prints time against
allocation order / virtual address.  Argument: budget in GiB (default 150)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

budget = float(sys.argv[1]) * (1 << 30) if len(sys.argv) > 1 else 150 * (1 << 30)
kind = sys.argv[2] if len(sys.argv) > 2 else "complex"
x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
plan = aa.Stft(1024, 256, output=kind)
nf = plan.num_frames(len(x))
st = torch.cuda.current_stream()
slab_bytes = nf * plan.cols * 4


def timeit(out, n=20, warm=5):
    for _ in range(warm):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


bufs, times = [], []
first = plan.alloc_rows(nf, candidates=1)[0]
timeit(first, 300, 300)
bufs.append(first)
while (len(bufs) + 1) * slab_bytes < budget:
    try:
        bufs.append(plan.alloc_rows(nf, candidates=1)[0])
    except aa.AaError:
        break
for b in bufs:
    times.append(timeit(b))
ts = sorted(times)
print(f"{len(bufs)} slabs of {slab_bytes / 1e6:.0f} MB ({len(bufs) * slab_bytes / (1 << 30):.1f} GiB): "
      f"min {ts[0]:.1f}  p10 {ts[len(ts) // 10]:.1f}  median {ts[len(ts) // 2]:.1f}  max {ts[-1]:.1f} us")
for i in range(0, len(bufs), 8):
    print(f"  slabs {i:3d}..: " + " ".join(f"{t:6.1f}" for t in times[i:i + 8]) + f"   first @ {bufs[i].data_ptr():#x}")
fastest = min(range(len(bufs)), key=lambda i: times[i])
print(json.dumps({"kind": kind, "slabs": len(bufs), "min_us": round(ts[0], 2), "median_us": round(ts[len(ts) // 2], 2),
                  "max_us": round(ts[-1], 2), "fast_below_100": sum(t < 100 for t in times), "fastest_index": fastest,
                  "again_fastest_us": round(timeit(bufs[fastest], 200, 50), 2)}))

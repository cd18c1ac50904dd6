"""Finds, in ONE process, a FAST and a SLOW output buffer for the complex-rows launch (slabs of the launch's output size are
allocated and timed one after the other - all kept, so each is another piece of physical memory - until one is >= 12 % faster
than the median), then issues 30 launches into the fast one and 30 into the slow one: under `rocprofv3 --pmc` these are the
LAST 60 dispatches of the stft kernel, fast first.  Prints a JSON line (times, slab count)."""
import os, sys, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

max_slabs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
plan = aa.Stft(1024, 256, output="complex")
nf = plan.num_frames(len(x))
st = torch.cuda.current_stream()


def timeit(out, n=4, warm=2):
    for _ in range(warm):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


bufs, times = [], []
fast = None
while len(bufs) < max_slabs:
    bufs.append(plan.alloc_rows(nf, candidates=1)[0])
    if len(bufs) == 1:
        timeit(bufs[0], 30, 30)
    times.append(timeit(bufs[-1]))
    # (an absolute margin: under a counter pass every launch carries the profiler's per-dispatch overhead, which dilutes a ratio)
    if len(times) >= 6 and statistics.median(times) - times[-1] > 14.0:
        # confirm against the slab nearest the median, 16 launches each (the profiler's per-dispatch overhead is noisy)
        ref = min(range(len(times) - 1), key=lambda i: abs(times[i] - statistics.median(times)))
        a, b = timeit(bufs[-1], 16, 2), timeit(bufs[ref], 16, 2)
        if b - a > 14.0:
            fast = len(bufs) - 1
            times[-1] = a
            break
med = statistics.median(times)
slow = min(range(len(times)), key=lambda i: abs(times[i] - med))
rec = {"slabs": len(bufs), "median_us": round(med, 2), "fast_index": fast, "slow_index": slow,
       "fast_us": None if fast is None else round(times[fast], 2), "slow_us": round(times[slow], 2)}
if fast is not None:
    for b in (bufs[fast], bufs[slow]):
        for _ in range(30):
            plan.execute_device(xd, 0, nf, out=b, stream=st.cuda_stream)
        torch.cuda.synchronize()
    rec["fast_ptr"], rec["slow_ptr"] = hex(bufs[fast].data_ptr()), hex(bufs[slow].data_ptr())
print(json.dumps(rec))

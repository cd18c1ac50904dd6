"""The complex-rows launch (process_forward's own output, 1024 / 256, 10 min) into K output buffers of ONE process, each a
plain allocation: launch time per buffer (the placement lottery of DESIGN.md section 5), then aa_rows_alloc's pick among K
fresh candidates.  `prof` as first argument: 40 launches into the fastest buffer, then 40 into the slowest (for a rocprofv3
--pmc pass: the dispatches come in that order).  Prints one JSON line at the end."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

prof = len(sys.argv) > 1 and sys.argv[1] == "prof"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kind = sys.argv[3] if len(sys.argv) > 3 else "complex"
lib = aa.load_library()
x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
plan = aa.Stft(1024, 256, output=kind)
nf = plan.num_frames(len(x))
st = torch.cuda.current_stream()


def timeit(out, n=200, warm=60):
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


t0 = time.time()
bufs = [plan.alloc_rows(nf, candidates=1)[0] for _ in range(K)]
timeit(bufs[0], 100, 300)                                  # clocks up
times = [timeit(b) for b in bufs]
again = [timeit(b, 100, 20) for b in bufs]
for i, (b, t, t2) in enumerate(zip(bufs, times, again)):
    print(f"plain allocation {i}: rows @ {b.data_ptr():#x}: {t:7.2f} us   (again {t2:7.2f})")
fast, slow = min(range(K), key=lambda i: times[i]), max(range(K), key=lambda i: times[i])
rec = {"kind": kind, "plain_us": [round(t, 2) for t in times], "fastest": fast, "slowest": slow}
if prof:
    for b in (bufs[fast], bufs[slow]):
        for _ in range(40):
            plan.execute_device(xd, 0, nf, out=b, stream=st.cuda_stream)
        torch.cuda.synchronize()
    rec["prof_order"] = "40 launches into the fastest buffer, then 40 into the slowest, after %d timing launches" % 0
else:
    del bufs
    t1 = time.time()
    picked, best, typical = plan.alloc_rows(nf, candidates=256)
    t_alloc = time.time() - t1
    chk = timeit(picked)
    print(f"aa_rows_alloc(up to 256 slabs): kept {best:.2f} us (median of the slabs tried {typical:.2f}) in {t_alloc:.2f} s; "
          f"measured again {chk:.2f} us")
    rec.update(alloc_kept_us=round(best, 2), alloc_typical_us=round(typical, 2), alloc_measured_us=round(chk, 2),
               alloc_seconds=round(t_alloc, 2))
print(json.dumps(rec))

"""Probe (diagnostic): launch time of the 1024/256 kernel against the number of persistent workgroups
(AA_GRID_SET), in one process - i.e. against the byte stride between the workgroups' write streams.
Run on a GPU box against a library built with -DAA_EXPERIMENTS=1 (the shipped one ignores the knob):
python tools/probe_bimodal_grid.py [kind]"""
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
st = torch.cuda.current_stream()


def timeit(n=300):
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


t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(50)
print(f"default grid: {timeit():.2f} us")
for g in (1024, 1023, 1021, 1016, 1008, 1000, 992, 960, 896, 1024):
    os.environ["AA_GRID_SET"] = str(g)
    print(f"grid {g:5d}: {timeit():.2f} us")

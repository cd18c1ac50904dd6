"""How long does the host take to issue one bench step, vs the kernel's own time?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch, numpy as np
import audio_analyzer_rs_amd as aa
from audio_analyzer_rs_amd import _lib
for kind in ("corpus", "noise"):
    n = 28_800_000
    x = aa.signals.synth_corpus(0, n) if kind == "corpus" else np.random.default_rng(0).uniform(-1, 1, n).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    plan = aa.Stft(1024, 256)
    nf = plan.num_frames(n)
    out = torch.empty((nf, 513), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib = _lib.load()
    def step_py():
        plan.execute_device(xd, 0, nf, out=out, stream=st)
    args = (plan._plan, ctypes.c_void_p(xd.data_ptr()), ctypes.c_int64(n), ctypes.c_int64(0), ctypes.c_int64(nf), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st))
    def step_raw():
        lib.aa_stft_execute_device(*args)
    for name, step in (("raw ctypes", step_raw), ("python wrapper", step_py), ("raw ctypes", step_raw), ("python wrapper", step_py)):
        for _ in range(20): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(300): step()
        t_issue = time.perf_counter() - t0
        e1.record(); torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{kind:7s} {name:15s}: host issue {t_issue/300*1e6:6.1f} us/step, wall {t_all/300*1e6:6.1f} us/step, events {e0.elapsed_time(e1)/300*1e3:6.1f} us/step")

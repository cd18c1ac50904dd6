"""Times the other BASELINE / reference geometries (not the headline bench line):
2048/512 tuner, 256/64 onset, log-mel, MFCC.  Prints frames/s and the fraction of the
8 TB/s HBM roofline on algorithmic bytes (hop*4 in + cols*4 out)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_analyzer_rs_amd as aa

if os.environ.get("AA_LIB_OVERRIDE"):      # variant builds (tools/build_variants.sh)
    aa._lib.LIB_PATH = os.path.abspath(os.environ["AA_LIB_OVERRIDE"])

x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
cases = [
    (1024, 256, "mag", {}), (1024, 256, "complex", {}), (2048, 512, "mag", {}), (2048, 512, "complex", {}),
    (256, 64, "mag", {}), (512, 128, "mag", {}), (128, 32, "mag", {}), (4096, 1024, "mag", {}),
    (2048, 512, "logmel", dict(n_mels=128)), (1024, 256, "logmel", dict(n_mels=128)),
    (1024, 256, "mfcc", dict(n_mels=128, n_mfcc=40)),
    (2048, 512, "logmel", dict(n_mels=128, gemm=1)), (1024, 256, "mfcc", dict(n_mels=128, n_mfcc=40, gemm=1)),
]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[2] in sys.argv[1:]]
for n, hop, outk, kw in cases:
    kw = dict(kw)
    gemm = kw.pop("gemm", 0)
    plan = aa.Stft(n, hop, output=outk, **kw)
    src = xd
    if gemm:      # the unfused power slab + MFMA GEMM filterbank: what a call on samples that are not 16-byte aligned runs
        src = torch.empty(xd.numel() + 1, device="cuda")[1:]
        src.copy_(xd)
    nf = plan.num_frames(src.shape[0])
    out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    reps = 20 if plan.kernel.startswith("wave") else 3
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        plan.execute_device(src, 0, nf, out=out, stream=st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.execute_device(src, 0, nf, out=out, stream=st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    bpf = hop * 4 + plan.cols * 4
    print(f"n_fft {n:5d} hop {hop:5d} {(outk + ('/gemm' if gemm else '')):11s} kernel {plan.kernel:10s} {nf:7d} frames  {ms*1e3:9.1f} us  "
          f"{nf/ms*1e3:10.4g} frames/s  {bpf} B/frame  {nf*bpf/ms/1e6/8000:6.3f} of 8 TB/s")

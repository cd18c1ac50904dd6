"""Diagnostic: does the fused MFCC / log-mel launch's time follow its OUTPUT buffer, its INPUT buffer or neither?  One process,
several input copies and output buffers allocated between spacers of odd sizes; every (input, output) pair timed (300 launches).
Usage (GPU box): python tools/probe_mel_placement.py [mfcc|logmel]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa

kind = sys.argv[1] if len(sys.argv) > 1 else "mfcc"
kw = dict(n_mels=128, n_mfcc=40) if kind == "mfcc" else dict(n_mels=128)
x = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
plan = aa.Stft(1024, 256, output=kind, **kw)
nf = plan.num_frames(x.shape[0])
st = torch.cuda.current_stream()
spacers, ins, outs = [], [x], []
for i in range(6):
    spacers.append(torch.empty(int(1e6 * (3 + 7 * i)) + 4 * i, dtype=torch.float32, device="cuda"))
    outs.append(torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda"))
    if i < 3:
        spacers.append(torch.empty(int(1e6 * (11 + 5 * i)), dtype=torch.float32, device="cuda"))
        ins.append(x.clone())
def t(xi, oi, n=300):
    for _ in range(100):
        plan.execute_device(ins[xi], 0, nf, out=outs[oi], stream=st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(ins[xi], 0, nf, out=outs[oi], stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(300):
    plan.execute_device(x, 0, nf, out=outs[0], stream=st.cuda_stream)
torch.cuda.synchronize()
print(kind, "us per launch; rows = input buffer, columns = output buffer")
print("      " + " ".join("out%-4d" % j for j in range(len(outs))))
for xi in range(len(ins)):
    print("in%-3d " % xi + " ".join("%6.1f " % t(xi, oi) for oi in range(len(outs))), "  in @%x" % ins[xi].data_ptr(), flush=True)
print("outs @", " ".join("%x" % o.data_ptr() for o in outs))

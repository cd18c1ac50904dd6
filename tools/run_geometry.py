"""Runs one geometry back to back (for rocprofv3 passes): python3 tools/run_geometry.py n_fft hop output reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa
n, hop, outk, reps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
plan = aa.Stft(n, hop, output=outk)
nf = plan.num_frames(xd.shape[0])
out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    plan.execute_device(xd, 0, nf, out=out, stream=st)
torch.cuda.synchronize()

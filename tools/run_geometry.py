"""Runs one geometry back to back (for rocprofv3 passes): python3 tools/run_geometry.py n_fft hop output reps [inputs]
inputs > 1: that many input / output buffer pairs alternating (samples from HBM, not the Infinity Cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa
if os.environ.get("AA_TOOL_LIB"):                    # another build of the library (experiments)
    aa._lib.LIB_PATH = os.path.abspath(os.environ["AA_TOOL_LIB"])
n, hop, outk, reps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
alt = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kw = dict(n_mels=128) if outk in ("logmel", "mfcc") else {}
if outk == "mfcc":
    kw["n_mfcc"] = 40
xs = [torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000, seed=12345 + 7 * i)).cuda() for i in range(alt)]
plan = aa.Stft(n, hop, output=outk, **kw)
nf = plan.num_frames(xs[0].shape[0])
outs = [torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda") for _ in range(alt)]
st = torch.cuda.current_stream().cuda_stream
for i in range(reps):
    plan.execute_device(xs[i % alt], 0, nf, out=outs[i % alt], stream=st)
torch.cuda.synchronize()

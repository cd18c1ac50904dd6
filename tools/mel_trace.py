import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_analyzer_rs_amd as aa
x = aa.signals.synth_corpus(0, 28_800_000)
xd = torch.from_numpy(x).cuda()
for n, hop, outk, kw in [(2048, 512, "logmel", dict(n_mels=128)), (1024, 256, "mfcc", dict(n_mels=128, n_mfcc=40))]:
    plan = aa.Stft(n, hop, output=outk, **kw)
    for _ in range(5):
        o = plan(xd)
    torch.cuda.synchronize()

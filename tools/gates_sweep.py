import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import audio_analyzer_rs_amd as aa
aa._lib.LIB_PATH = os.path.abspath(sys.argv[1])
xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
mags = aa.Stft(256, 64, output="mag")(xd)
nf, nb = mags.shape
gf = torch.full((nf,), float(10.0 ** (-60 / 20.0) * nb / 2.0), device="cuda")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for ns in [int(v) for v in os.environ.get("AA_SWEEP", "1,16,64,128,256,512,1024,2048,4096,8192,16384").split(",")]:
    offs = torch.tensor(np.linspace(0, nf, ns + 1).astype(np.int64), device="cuda")
    flux, energy = aa.onset_flux(mags, offsets=offs)
    print(os.path.basename(sys.argv[1]), ns, "streams:", round(timed(lambda: aa.onset_gates(mags, flux, energy, gf, offsets=offs)), 4), "ms", flush=True)

"""Diagnostic: the per-process slow mode of the 1024-point mel launches in bench.py's context.  Runs `bench.py --skip-cpu`
(short) in fresh processes and prints the headline, MFCC and 1024 log-mel times of each; then, still in that process, times the
MFCC launch into six NEW output buffers and from a new copy of the input (does the mode belong to the process or to a buffer?).
Usage (GPU box): python tools/probe_mfcc_lottery.py [processes=6] [library]
Findings (r4, profiles/r04_bench_repeats.txt): about one process in four runs BOTH 1024 mel launches ~10 % slower (MFCC 90-101 us
against 83, log-mel 81-84 against 77) with plain as with nontemporal coefficient stores; the headline launch and the 2048 mel
launch of the same process are unaffected; grid size, LDS and register counts of the dispatches are the same (rocprofv3)."""
import os, subprocess, sys

CHILD = r"""
import os, sys, json, io, contextlib
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--skip-cpu', '--steps', '50', '--sustained-s', '0.05', '--config-steps', '100']
import torch
import audio_analyzer_rs_amd as aa
if os.environ.get('AA_LIB_FOR_BENCH'):
    aa._lib.LIB_PATH = os.environ['AA_LIB_FOR_BENCH']
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
c = d['configs']
plan = aa.Stft(1024, 256, output='mfcc', n_mels=128, n_mfcc=40)
x = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
nf = plan.num_frames(x.shape[0])
st = torch.cuda.current_stream()
def t(xi, out, n=200):
    for _ in range(300):
        plan.execute_device(xi, 0, nf, out=out, stream=st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(xi, 0, nf, out=out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
after = [t(x, torch.empty((nf, 40), device='cuda')) for _ in range(6)]
print('head %.1f | bench: mfcc %.1f logmel %.1f | afterwards, new plan / input / six outputs: %s' % (
    d['roofline']['kernel_ms'] * 1e3, c['cfg4_mfcc_1024']['ms'] * 1e3, c['logmel_1024_256']['ms'] * 1e3,
    ' '.join('%.1f' % v for v in after)))
"""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
env = dict(os.environ)
if len(sys.argv) > 2:
    env["AA_LIB_FOR_BENCH"] = os.path.abspath(sys.argv[2])
for rep in range(n):
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("process %d: %s" % (rep, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)

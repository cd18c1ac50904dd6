"""Diagnostic: the per-process slow mode of the 1024-point mel launches in bench.py's context.  Runs `bench.py --skip-cpu`
(short) in fresh processes and prints the headline, MFCC and 1024 log-mel times of each; then, still in that process, times the
MFCC launch into six NEW output buffers and from a new copy of the input (does the mode belong to the process or to a buffer?).
Usage (GPU box): python tools/probe_mfcc_lottery.py [processes=6] [library]
Findings (r4, profiles/r04_bench_repeats.txt): the "slow mode" (MFCC 90-101 us against 83, log-mel 81-84 against 77, about one
bench process in four) followed whatever was timed right behind seconds of multi-threaded numpy work on the host - its worker
threads keep spinning, the launching thread loses its core now and then and the GPU queue runs dry.  Not the GPU, not a buffer,
not the plan: the same objects re-timed a second later ran at their usual speed.  bench.py was reordered accordingly."""
import os, subprocess, sys

CHILD = r"""
import os, sys, json, io, contextlib
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--skip-cpu', '--steps', '50', '--sustained-s', '0.05', '--config-steps', '100']
import torch
import audio_analyzer_rs_amd as aa
if os.environ.get('AA_LIB_FOR_BENCH'):
    aa._lib.LIB_PATH = os.environ['AA_LIB_FOR_BENCH']
# keep the bench's own MFCC plan, input and output alive: the last full-size MFCC call it makes
kept = {}
orig = aa.Stft.execute_device
def spy(self, samples, frame_begin=0, frame_end=None, out=None, stream=None):
    if self.output == 'mfcc' and out is not None and samples.numel() == 28_800_000:
        kept.update(plan=self, x=samples, out=out, stream=stream)
    return orig(self, samples, frame_begin, frame_end, out, stream)
aa.Stft.execute_device = spy
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
aa.Stft.execute_device = orig
d = json.loads(buf.getvalue().strip().splitlines()[-1])
c = d['configs']
st = torch.cuda.current_stream()
def t(plan, xi, out, stream=None, n=200):
    s = stream if stream is not None else st.cuda_stream
    for _ in range(300):
        plan.execute_device(xi, 0, out.shape[0], out=out, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        plan.execute_device(xi, 0, out.shape[0], out=out, stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
plan, x, out = kept['plan'], kept['x'], kept['out']
new_plan = aa.Stft(1024, 256, output='mfcc', sample_rate=48000, n_mels=128, n_mfcc=40)
new_x = x.clone()
new_out = torch.empty_like(out)
print('head %.1f | bench: mfcc %.1f logmel %.1f | its own plan, input, output again %.1f | new output %.1f | new input %.1f | new plan %.1f | all new %.1f'
      ' | x @%x out @%x new_out @%x' % (
    d['roofline']['kernel_ms'] * 1e3, c['cfg4_mfcc_1024']['ms'] * 1e3, c['logmel_1024_256']['ms'] * 1e3,
    t(plan, x, out), t(plan, x, new_out), t(plan, new_x, out), t(new_plan, x, out), t(new_plan, new_x, new_out),
    x.data_ptr(), out.data_ptr(), new_out.data_ptr()))
"""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
env = dict(os.environ)
if len(sys.argv) > 2:
    env["AA_LIB_FOR_BENCH"] = os.path.abspath(sys.argv[2])
for rep in range(n):
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("process %d: %s" % (rep, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)

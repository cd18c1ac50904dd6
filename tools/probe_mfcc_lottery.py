"""Diagnostic: the per-process slow mode of the mel launches in bench.py's context.  Runs `bench.py --skip-cpu` (short) in five
fresh processes per library and prints the headline, MFCC and 1024 log-mel times of each.
Usage (GPU box): python tools/probe_mfcc_lottery.py audio_analyzer_rs_amd/libaa_stft.so [another build ...]
Finding (r4, profiles/r04_bench_repeats.txt): about one process in five runs BOTH mel launches ~10 % slower (MFCC 95-101 us
against 83, log-mel 81-84 against 77) with plain as with nontemporal coefficient stores; the headline launch of the same process
is unaffected."""
import os, sys, json, subprocess
libs = sys.argv[1:]
for rep in range(5):
    for lib in libs:
        env = dict(os.environ, AA_LIB_FOR_BENCH=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", """
import os, sys, json, io, contextlib
sys.argv = ['bench.py', '--skip-cpu', '--steps', '50', '--sustained-s', '0.05', '--config-steps', '100']
import audio_analyzer_rs_amd as aa
aa._lib.LIB_PATH = os.environ['AA_LIB_FOR_BENCH']
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
c = d['configs']
print('%-14s head %.1f mfcc %.2f logmel %.2f' % (os.path.basename(os.environ['AA_LIB_FOR_BENCH']), d['roofline']['kernel_ms']*1e3, c['cfg4_mfcc_1024']['ms']*1e3, c['logmel_1024_256']['ms']*1e3))
"""], env=env, capture_output=True, text=True)
        print((out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)

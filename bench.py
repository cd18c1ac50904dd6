"""Benchmark of the STFT hot path on MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 — BASELINE configs[1], the configuration the metric is quoted on: one "step" = one launch of
the fused kernel (frame + periodic Hann + 1024-pt real FFT + |X|, hop 256) over 10 min of synthetic
48 kHz mono f32 (112 497 frames), input and output resident in HBM.  The JSON line also carries
  roofline      the kernel's algorithmic bytes (hop*4 in + 513*4 out = 3076 B/frame)
                over its average launch duration (HIP events on the launch stream) against 8 TB/s;
                `copy_ceiling`: what a plain device copy and the launch's 1 : 2 read : write mix of the
                same byte count reach in this very run (the measured ceiling next to the nominal one);
  sustained_*   >= 2 s of back-to-back launches of the same kernel between one pair of events;
  post_check    >= 64 rows of the slab the TIMED launches wrote, compared with the float64 truth
                (first / last / ragged-tail / random rows); a failure exits non-zero;
  configs       the other BASELINE configurations that fit one GPU, each timed the same way:
                the reference's own geometries 2048/512 and 256/64 (stft.rs:169-170,
                onset.rs:122-123), process_forward's complex output, configs[2] (2048-pt + 128
                log-mel on stereo, per channel and downmix-first as mod.rs:789-793 does) and
                configs[4] (40-coef MFCC on a sine and a chirp, with the error vs the truth);
  cpu_baseline  the CPU restatement of the reference path (oracle/, kind "port": the Rust crate
                cannot be built here and rustfft's AVX butterflies would be several times faster
                than this scalar radix-2 port) timed on this host, plus configs[0]'s stand-in.

N > 1 — BASELINE configs[3]: the 1 h / 172.8 M-sample / 674 997-frame corpus, frame-range sharded
over the N ranks (strong scaling; one rank per GPU, launched by torch.distributed.run, or spawned
from here when WORLD_SIZE is unset).  Every rank synthesises its own sample range.  `value` is
(i) compute only, outputs stay sharded, no data-path collective; `e2e` is (ii) scatter of the
samples from rank 0 + compute + gather of the rows to rank 0 over RCCL (grouped send/recv): the three
legs one after the other (`serial_total_ms`) and overlapped - every rank's range in pieces, scatter
k + 1 || compute k || gather k - 1 (`total_ms`); `e2e_cabi` the same on the C-ABI collectives.

Rank 0 prints ONE JSON line (kept compact); the full record goes to gpurun_out/bench_detail_nN.json.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FFT, HOP, SR = 1024, 256, 48000
SAMPLES_10MIN = 600 * SR                     # configs[1]
SAMPLES_1H = 3600 * SR                       # configs[3]
BYTES_PER_FRAME = HOP * 4 + (N_FFT // 2 + 1) * 4   # 3076: algorithmic bytes
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3                 # MI355X_MICROARCH.md: dense f32 matrix peak
TOL_VS_TRUTH = 5e-6                          # tests/conftest.py (implies <= 1e-5 vs the Rust CPU path)


def r4(v):
    return float(f"{v:.4g}")


# --------------------------------------------------------------------------- checks --
def check_rows(x, got_rows, frame_ids, n_fft, hop, kind):
    """max-normalised error (per frame max|out-ref| / max|ref|) of the given rows
    against the float64 truth of oracle/truth.py — the checker, never the thing measured."""
    from oracle import truth

    w = truth.hann_f32(n_fft)
    idx = np.asarray(frame_ids)[:, None] * hop + np.arange(n_fft)[None, :]
    fr = (x[idx] * w[None, :]).astype(np.float32).astype(np.float64)
    spec = np.fft.rfft(fr, axis=-1)
    ref = spec if kind == "complex" else np.abs(spec)
    got = np.asarray(got_rows)
    if kind == "complex":
        got = got.astype(np.complex128)
        num = np.max(np.abs(got - ref), axis=-1)
    else:
        num = np.max(np.abs(got.astype(np.float64) - ref), axis=-1)
    den = np.max(np.abs(ref), axis=-1)
    return float(np.max(num / np.where(den == 0, 1.0, den)))


def pick_rows(nf, n=64, seed=0):
    """first 16, last 16 (the ragged tail of the last chunk) and random rows."""
    rng = np.random.default_rng(seed)
    ids = set(range(min(16, nf))) | set(range(max(0, nf - 16), nf))
    while len(ids) < min(n, nf):
        ids.add(int(rng.integers(0, nf)))
    return np.array(sorted(ids), dtype=np.int64)


# ----------------------------------------------------------------------- CPU baseline --
def cpu_baseline(x, min_seconds=8.0, max_reps=20):
    """The CPU restatement (oracle/) on rank 0's 10-min buffer: single thread, then all cores;
    then configs[0]'s stand-in (60 s of 16-bit PCM as recorder.rs:81-85 writes it, decoded
    i16/32768, 1024/256 periodic Hann, single thread)."""
    import oracle
    import audio_analyzer_rs_amd as aa

    oracle.build_oracle()
    nf = oracle.num_frames(len(x), N_FFT, HOP)
    w = oracle.hann_window(N_FFT)
    oracle.stft(x[:N_FFT + 255 * HOP], N_FFT, HOP, kind="mag", window=w)      # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.stft(x, N_FFT, HOP, kind="mag", window=w)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or reps >= max_reps:
            break
    single = reps * nf / dt

    from concurrent.futures import ThreadPoolExecutor

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    bounds = [nf * i // cores for i in range(cores + 1)]

    def work(i):
        oracle.stft(x, N_FFT, HOP, kind="mag", window=w, f0=bounds[i], f1=bounds[i + 1])

    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))                                     # warm
        t1 = time.perf_counter()
        mreps = 0
        while True:
            list(ex.map(work, range(cores)))
            mreps += 1
            mdt = time.perf_counter() - t1
            if mdt >= min_seconds / 2 or mreps >= max_reps:
                break
    multi = mreps * nf / mdt

    # configs[0] stand-in: test_output.wav is stripped from the reference (.MISSING_LARGE_BLOBS)
    pcm = aa.wav.decode_pcm16(aa.wav.encode_pcm16(x[:60 * SR]))
    nf0 = oracle.num_frames(len(pcm), N_FFT, HOP)
    r0, t2 = 0, time.perf_counter()
    while True:
        oracle.stft(pcm, N_FFT, HOP, kind="mag", window=w)
        r0 += 1
        d0 = time.perf_counter() - t2
        if d0 >= min_seconds / 2 or r0 >= 200:
            break
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": r4(single), "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"10-min buffer x {reps} passes, {dt:.1f} s, 1 thread; scalar radix-2 C port "
                  "(gcc -O3 -mavx2 -ffp-contract=off), not the rustfft binary",
        "all_cores": {"value": r4(multi), "cores": cores},
        "config0": {"value": r4(r0 * nf0 / d0), "unit": "frames/s", "cores": 1,
                    "sample": f"60 s PCM16 stand-in ({nf0} frames) x {r0} passes, {d0:.1f} s"},
        "cpu_model": cpu_model,
    }


def load_traffic():
    """HBM bytes per launch of the headline kernel from the committed PMC passes (NOT measured in
    this run): (bytes, source file)."""
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                return json.load(f).get("hbm_bytes_per_launch"), "profiles/" + name
        except (OSError, ValueError):
            continue
    return None, None


# ------------------------------------------------------------------------ GPU timing --
class GpuTimer:
    """K launches bracketed by HIP events recorded on the stream the kernels are launched on."""

    def __init__(self, torch, dev):
        self.torch, self.dev = torch, dev
        self.stream = torch.cuda.current_stream(dev)

    def ramp(self, fn, seconds=0.25):
        # Device bring-up (not a step): a cold MI355X needs ~30 ms of back-to-back launches to reach
        # its steady clock / page-table state (tools/launch_overhead.py).
        t = time.perf_counter()
        while time.perf_counter() - t < seconds:
            for _ in range(50):
                fn()
            self.torch.cuda.synchronize(self.dev)

    def settle(self, fn, min_seconds=0.1, max_seconds=1.5, window=100, tol=0.01):
        """Ramp until two consecutive windows of `window` launches agree within `tol` (and at least `min_seconds`).  The
        entries of `configs` are timed right behind host-only phases (truth computations: the GPU idles for seconds); about
        one process in four then ran the mel launches 10 % slow for longer than a fixed 0.15 s ramp - and at its usual speed
        when timed again a second later (tools/probe_mfcc_lottery.py, profiles/r04_bench_repeats.txt).  Returns seconds spent."""
        torch = self.torch
        t0 = time.perf_counter()
        prev, agree = None, 0
        while True:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            for _ in range(window):
                fn()
            e1.record(self.stream)
            torch.cuda.synchronize(self.dev)
            cur = e0.elapsed_time(e1)
            agree = agree + 1 if prev is not None and abs(cur - prev) <= tol * prev else 0
            prev = cur
            spent = time.perf_counter() - t0
            if (agree >= 2 and spent >= min_seconds) or spent >= max_seconds:
                return spent

    def time(self, fn, steps, warmup):
        torch = self.torch
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(self.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(self.stream)
        for _ in range(steps):
            fn()
        e1.record(self.stream)
        while not e1.query():                      # poll: a blocking wait adds the host's wake-up latency to K short steps
            pass
        torch.cuda.synchronize(self.dev)
        wall = time.perf_counter() - t0
        return e0.elapsed_time(e1) / steps, wall


def sustained_leg(timer, step, kernel_ms, seconds=2.0):
    """>= `seconds` of back-to-back headline launches between one pair of HIP events: the power-limited steady state
    (the K-step figure above can still ride the boost clock: DESIGN.md section 5), long enough for an SMI sampler to see."""
    n = max(int(seconds / (kernel_ms * 1e-3)) + 1, 100)
    ms, wall = timer.time(step, n, 0)
    return {"launches": n, "seconds": r4(wall), "ms_per_step": ms}


def copy_ceiling(aa, torch, dev, timer, n_in, n_out):
    """What this box's memory system delivers for the launch's byte count with NO arithmetic, in the product's own access
    idiom (aa_diag_stream_copy_device: 16-byte loads, 16-byte nontemporal stores, persistent workgroups on contiguous runs):
    (i) a plain copy of (in + out) / 2 floats (reads as many bytes as it writes), (ii) the launch's own 1 : 2 read :
    write mix (every input piece stored twice).  HIP events over 50 runs after a ramp, on the launch stream.
    SURVEY.md section 8d: "report both nominal and measured-copy ceilings".
    r5: a ceiling must not sit below what the product moves.  Rounds 3-4 used torch's copy_ into torch allocations
    (5.3-5.4 TB/s where the product's complex-row launch moved 6.4): the write rate of this chip depends on the PHYSICAL
    memory behind the destination (one 462-MB slab in twenty takes write streams ~25 % faster), so
    the destinations here are placed by measurement (aa_rows_alloc, with the complex plan's launch as the probe) and the
    same copies into ordinary allocations are reported next to them."""
    lib = aa.load_library()
    st = timer.stream.cuda_stream
    half = ((n_in + n_out) // 2) & ~3
    nin4 = n_in & ~3
    src = torch.empty(half, dtype=torch.float32, device=dev).normal_()
    a = torch.empty(nin4, dtype=torch.float32, device=dev).normal_()
    probe = aa.Stft(N_FFT, HOP, output="complex")
    rows_needed = -(-max(half, 4 * nin4) // probe.cols)
    out = {}
    for tag in ("placed", "plain"):
        if tag == "placed":
            slab, kept_us, typical_us = probe.alloc_rows(rows_needed, candidates=128)
            slab = slab.view(-1)
            out["placement_probe_us"] = [r4(kept_us), r4(typical_us)]
        else:
            slab = torch.empty(rows_needed * probe.cols, dtype=torch.float32, device=dev)
        dst, b, b4 = slab[:half], slab[:2 * nin4], slab[:4 * nin4]

        def copy():
            aa._lib.check(lib.aa_diag_stream_copy_device(src.data_ptr(), dst.data_ptr(), half, 1, st))

        def mix():
            aa._lib.check(lib.aa_diag_stream_copy_device(a.data_ptr(), b.data_ptr(), nin4, 2, st))

        def mix4():
            aa._lib.check(lib.aa_diag_stream_copy_device(a.data_ptr(), b4.data_ptr(), nin4, 4, st))

        timer.ramp(copy, 0.1)
        ms_copy, _ = timer.time(copy, 50, 5)
        torch.cuda.synchronize(dev)
        ok = bool(torch.equal(dst, src))
        timer.ramp(mix, 0.1)
        ms_mix, _ = timer.time(mix, 50, 5)
        torch.cuda.synchronize(dev)
        ok = ok and bool(torch.equal(b[:nin4], a)) and bool(torch.equal(b[nin4:], a))
        timer.ramp(mix4, 0.1)
        ms_mix4, _ = timer.time(mix4, 50, 5)
        torch.cuda.synchronize(dev)
        ok = ok and all(bool(torch.equal(b4[w * nin4:(w + 1) * nin4], a)) for w in range(4))
        out[tag] = {"copy_GBps": 2 * half * 4 / (ms_copy * 1e-3) / 1e9, "mix_1r2w_GBps": 3 * nin4 * 4 / (ms_mix * 1e-3) / 1e9,
                    "mix_1r4w_GBps": 5 * nin4 * 4 / (ms_mix4 * 1e-3) / 1e9, "copied_ok": ok}
        del dst, b, b4, slab
    del src, a, probe
    return {"copy_GBps": out["placed"]["copy_GBps"], "mix_1r2w_GBps": out["placed"]["mix_1r2w_GBps"],
            "mix_1r4w_GBps": out["placed"]["mix_1r4w_GBps"],
            "copied_ok": out["placed"]["copied_ok"] and out["plain"]["copied_ok"],
            "plain_alloc": {"copy_GBps": r4(out["plain"]["copy_GBps"]), "mix_1r2w_GBps": r4(out["plain"]["mix_1r2w_GBps"]),
                            "mix_1r4w_GBps": r4(out["plain"]["mix_1r4w_GBps"])},
            "placement_probe_us": out["placement_probe_us"],
            "how": "aa_diag_stream_copy_device: dwordx4 loads / nontemporal dwordx4 stores, (in + out) / 2 floats copied; "
                   "the input floats stored twice; destinations placed by aa_rows_alloc (plain_alloc: torch allocations)"}


def other_configs(aa, torch, dev, timer, x, xd, steps, warmup):
    """The BASELINE configurations beside the headline, each timed like it (HIP events over
    `steps` launches after `warmup`), with algorithmic bytes per frame and the HBM fraction."""
    from oracle import truth
    import oracle

    st = timer.stream.cuda_stream
    out = {}

    def geometry(key, n_fft, hop, kind, bytes_per_frame, xin=xd, xhost=x, **kw):
        placed = kw.pop("placed", False)
        plan = aa.Stft(n_fft, hop, output=kind, **kw)
        nf = plan.num_frames(xin.shape[0])
        extra = {}
        if placed:
            # the rows placed by measurement (aa_rows_alloc: the complex launch's time depends on the physical memory
            # behind its output) - the product's recommended allocation for this output kind; the
            # same launch into an ordinary allocation is timed next to it
            plain = torch.empty((nf, plan.cols), dtype=torch.float32, device=dev)
            timer.settle(lambda: plan.execute_device(xin, 0, nf, out=plain, stream=st))
            ms_plain, _ = timer.time(lambda: plan.execute_device(xin, 0, nf, out=plain, stream=st), steps, warmup)
            del plain
            t_alloc = time.perf_counter()
            buf, kept_us, typical_us = plan.alloc_rows(nf, candidates=128)
            extra = {"rows": "aa_rows_alloc", "ms_plain_alloc": r4(ms_plain), "alloc_s": r4(time.perf_counter() - t_alloc),
                     "alloc_kept_us": r4(kept_us), "alloc_typical_us": r4(typical_us),
                     "frac_plain_alloc": r4(bytes_per_frame * nf / (ms_plain * 1e-3) / 1e9 / HBM_PEAK_GBS)}
            buf = buf.view(nf, plan.cols)
        else:
            buf = torch.empty((nf, plan.cols), dtype=torch.float32, device=dev)

        def step():
            plan.execute_device(xin, 0, nf, out=buf, stream=st)

        timer.settle(step)
        ms, _ = timer.time(step, steps, warmup)
        rec = {"kernel": plan.kernel, "frames": nf, "ms": r4(ms), "B_per_frame": bytes_per_frame,
               "frac": r4(bytes_per_frame * nf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS)}
        rec.update(extra)
        if kind in ("mag", "complex"):
            ids = pick_rows(nf, 64, seed=n_fft)
            rows = buf[torch.from_numpy(ids).to(dev)].cpu().numpy()
            if kind == "complex":
                rows = rows.reshape(len(ids), -1, 2)
                rows = rows[..., 0] + 1j * rows[..., 1]
            rec["err"] = r4(check_rows(xhost, rows, ids, n_fft, hop, kind))
        out[key] = rec
        return plan, buf, rec

    # the reference's own geometries and process_forward's real output format
    geometry("mag_2048_512", 2048, 512, "mag", 512 * 4 + 1025 * 4)
    geometry("mag_256_64", 256, 64, "mag", 64 * 4 + 129 * 4)
    geometry("complex_1024_256", 1024, 256, "complex", 256 * 4 + 513 * 8, placed=True)

    # configs[2]: 2048-pt + 128 log-mel on synthetic stereo (two channels, seeds 12345 / 54321)
    n_mels = 128
    xr = aa.signals.synth_corpus(0, SAMPLES_10MIN, seed=54321)
    xrd = torch.from_numpy(xr).to(dev)
    plan = aa.Stft(2048, 512, output="logmel", sample_rate=SR, n_mels=n_mels)
    nf2 = plan.num_frames(SAMPLES_10MIN)
    lm = torch.empty((2, nf2, n_mels), dtype=torch.float32, device=dev)

    def stereo_step():
        plan.execute_device(xd, 0, nf2, out=lm[0], stream=st)
        plan.execute_device(xrd, 0, nf2, out=lm[1], stream=st)

    timer.settle(stereo_step)
    ms_two, _ = timer.time(stereo_step, steps, warmup)
    # both channels in ONE launch (aa_stft_execute_channels_device: planar channels, the persistent
    # workgroups split between them; rows bit-identical to the two launches above)
    planar = torch.stack([xd, xrd]).contiguous()

    def stereo_one_launch():
        plan.execute_channels_device(planar, out=lm, stream=st)

    timer.settle(stereo_one_launch)
    ms, _ = timer.time(stereo_one_launch, steps, warmup)
    del planar
    b2 = 512 * 4 + n_mels * 4
    ids = pick_rows(nf2, 32, seed=3)
    P = np.abs(truth.stft_complex(np.concatenate([xr[i * 512:i * 512 + 2048] for i in ids]), 2048, 2048,
                                  oracle.hann_window(2048))) ** 2
    M = np.maximum(P @ truth.mel_filterbank(SR, 2048, n_mels).T, 1e-10)
    got = 10.0 ** (lm[1][torch.from_numpy(ids).to(dev)].cpu().numpy().astype(np.float64) / 10.0)
    out["cfg2_logmel_2048_stereo"] = {
        "kernel": plan.kernel, "frames": 2 * nf2, "ms": r4(ms), "ms_two_launches": r4(ms_two), "B_per_frame": b2,
        "frac": r4(b2 * 2 * nf2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
        "err": r4(float(truth.max_normalised_error(got, M).max()))}
    # reference semantics: (L+R)/2 at capture (mod.rs:789-793), then ONE mono STFT
    inter = torch.stack([xd, xrd], dim=1).contiguous()            # interleaved L R L R ..
    mono = torch.empty(SAMPLES_10MIN, dtype=torch.float32, device=dev)

    def downmix_step():
        aa._lib.check(aa._lib.load().aa_downmix_device(inter.data_ptr(), SAMPLES_10MIN, 2, mono.data_ptr(), st))
        plan.execute_device(mono, 0, nf2, out=lm[0], stream=st)

    timer.settle(downmix_step)
    ms, _ = timer.time(downmix_step, steps, warmup)
    bd = 12 * 512 + b2                                            # downmix (2+1)*4 B per sample + log-mel
    out["cfg2_downmix_first"] = {"frames": nf2, "ms": r4(ms), "B_per_frame": bd,
                                 "frac": r4(bd * nf2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS)}
    del inter, mono, lm, xrd

    # configs[4]: 40-coef MFCC (STFT -> mel -> log -> DCT-II) on a sine and a chirp (10 s, -60 dB dither as
    # tests/test_gpu_parity.py::test_mfcc_sine_and_chirp), error vs the float64 definition; throughput on the
    # 10-min corpus.  MFMA fraction = useful DCT flops (2*128*40 per frame) over the f32 matrix peak (the
    # kernel is bound by its FFT / mel VALU work, the matrix cores idle most of the launch).
    n_mfcc = 40
    mplan = aa.Stft(N_FFT, HOP, output="mfcc", sample_rate=SR, n_mels=n_mels, n_mfcc=n_mfcc)
    dct = truth.dct2_ortho(n_mfcc, n_mels)
    rng = np.random.default_rng(1)
    dither = (1e-3 * rng.standard_normal(10 * SR)).astype(np.float32)
    # (the float64 references of the two signals are computed BEHIND the timed launches below: seconds of host-only work
    # in front of a timing window leave the GPU in a state in which, in about one process in four, the mel launches run
    # 10 % slow for a second or so - tools/probe_mfcc_lottery.py, profiles/r04_bench_repeats.txt)
    checks = []
    for sig in (aa.signals.sine(440.0, 10 * SR), aa.signals.chirp(50.0, 12000.0, 10 * SR)):
        s = (sig + dither).astype(np.float32)
        checks.append((s, mplan(torch.from_numpy(s).to(dev)).cpu().numpy()))
    nf = mplan.num_frames(SAMPLES_10MIN)
    mbuf = torch.empty((nf, n_mfcc), dtype=torch.float32, device=dev)

    def mfcc_step():
        mplan.execute_device(xd, 0, nf, out=mbuf, stream=st)

    mfcc_settle = timer.settle(mfcc_step)
    ms, _ = timer.time(mfcc_step, steps, warmup)
    bm = HOP * 4 + n_mfcc * 4
    out["cfg4_mfcc_1024"] = {
        "kernel": mplan.kernel, "frames": nf, "ms": r4(ms), "B_per_frame": bm,
        "frac": r4(bm * nf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
        "mfma_frac": r4(2.0 * n_mels * n_mfcc * nf / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS),
        "settled_after_s": r4(mfcc_settle)}
    lplan = aa.Stft(N_FFT, HOP, output="logmel", sample_rate=SR, n_mels=n_mels)
    lbuf = torch.empty((nf, n_mels), dtype=torch.float32, device=dev)

    def logmel_step():
        lplan.execute_device(xd, 0, nf, out=lbuf, stream=st)

    timer.settle(logmel_step)
    ms, _ = timer.time(logmel_step, steps, warmup)
    bl = HOP * 4 + n_mels * 4
    out["logmel_1024_256"] = {"kernel": lplan.kernel, "frames": nf, "ms": r4(ms), "B_per_frame": bl,
                              "frac": r4(bl * nf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS)}
    del mbuf, lbuf
    out["next_rows"] = next_rows(aa, torch, dev, timer, xd)
    errs = []
    for s, got in checks:
        P = np.abs(truth.stft_complex(s, N_FFT, HOP, oracle.hann_window(N_FFT))) ** 2
        L = 10.0 * np.log10(np.maximum(P @ truth.mel_filterbank(SR, N_FFT, n_mels).T, 1e-10))
        errs.append(float(truth.max_normalised_error(got, L @ dct.T).max()))
    out["cfg4_mfcc_1024"].update(err_sine=r4(errs[0]), err_chirp=r4(errs[1]))
    return out


def next_rows(aa, torch, dev, timer, xd):
    """SURVEY.md section 8(f) rows behind the STFT, timed on the 10-min corpus cut into independent streams (their
    recurrences run over frames: one wave / one lane per bin per stream, so the parallelism is across streams):
    onset flux + gates on the 256/64 magnitudes (onset.rs:261-371), tuner noise floor + pitch extraction on the
    2048/512 magnitudes (stft.rs:326-367, 443-620).  Parity of each is the GPU test suite's business."""
    st = timer.stream.cuda_stream
    rec = {}
    try:
        mags = aa.Stft(256, 64, output="mag")(xd)
        torch.cuda.synchronize(dev)
        nfo, nbo = mags.shape
        gf = torch.full((nfo,), float(10.0 ** (-60 / 20.0) * nbo / 2.0), device=dev)
        for ns in (1024,):
            offs = torch.tensor(np.linspace(0, nfo, ns + 1).astype(np.int64), device=dev)
            flux, energy = aa.onset_flux(mags, offsets=offs, stream=st)
            torch.cuda.synchronize(dev)

            def onset_chain():
                f, e = aa.onset_flux(mags, offsets=offs, stream=st)
                aa.onset_gates(mags, f, e, gf, offsets=offs, stream=st)

            ms, _ = timer.time(onset_chain, 20, 3)
            ms_flux, _ = timer.time(lambda: aa.onset_flux(mags, offsets=offs, stream=st), 20, 3)
            # algorithmic bytes: the flux reads every row once and writes 8 B per frame; the gates read every row once,
            # 13 B of per-frame inputs (flux, energy, global floor, mask) and write 21 B
            bpf = 2 * nbo * 4 + 8 + 12 + 21
            rec["onset_flux_gates_256_64"] = {"frames": nfo, "streams": ns, "ms": r4(ms), "flux_ms": r4(ms_flux),
                                              "frames_per_s": r4(nfo / (ms * 1e-3)), "B_per_frame": bpf,
                                              "frac": r4(nfo * bpf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
                                              "flux_frac": r4(nfo * (nbo * 4 + 8) / (ms_flux * 1e-3) / 1e9 / HBM_PEAK_GBS)}
        del mags, flux, energy, gf
        mags = aa.Stft(2048, 512, output="mag")(xd)
        torch.cuda.synchronize(dev)
        nft, nbt = mags.shape
        gft = torch.full((nft,), 0.5, device=dev)
        offs = torch.tensor(np.linspace(0, nft, 257).astype(np.int64), device=dev)
        eff, _ = aa.tuner_floor(mags, gft, offsets=offs, stream=st)

        def tuner_chain():
            e, _ = aa.tuner_floor(mags, gft, offsets=offs, stream=st)
            aa.tuner_pitches(mags, e, SR / 2048, stream=st)

        ms, _ = timer.time(tuner_chain, 20, 3)
        ms_p, _ = timer.time(lambda: aa.tuner_pitches(mags, eff, SR / 2048, stream=st), 20, 3)
        # the floor reads a row and writes the effective floor; the pitches read both; 68 B of notes per frame
        bpf = 4 * nbt * 4 + 68
        rec["tuner_floor_pitches_2048_512"] = {"frames": nft, "streams": 256, "ms": r4(ms), "pitches_ms": r4(ms_p),
                                               "frames_per_s": r4(nft / (ms * 1e-3)), "B_per_frame": bpf,
                                               "frac": r4(nft * bpf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
                                               "pitches_frac": r4(nft * (2 * nbt * 4 + 68) / (ms_p * 1e-3) / 1e9 / HBM_PEAK_GBS)}
    except Exception as ex:                                   # noqa: BLE001 (reported, never fatal for the headline line)
        rec["error"] = f"{type(ex).__name__}: {ex}"
    return rec


# --------------------------------------------------------------------------- N == 1 --
def run_single(args):
    import torch

    import audio_analyzer_rs_amd as aa

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    x = aa.signals.synth_corpus(0, SAMPLES_10MIN)
    xd = torch.from_numpy(x).to(dev)
    plan = aa.Stft(N_FFT, HOP, output="mag")
    assert plan.kernel == "wave1024", plan.kernel
    nf = plan.num_frames(xd.shape[0])
    out = torch.empty((nf, plan.cols), dtype=torch.float32, device=dev)
    timer = GpuTimer(torch, dev)
    st = timer.stream.cuda_stream

    def step():
        plan.execute_device(xd, 0, nf, out=out, stream=st)

    out.fill_(float("nan"))                          # the timed launches must write every checked row
    timer.ramp(step)
    kernel_ms, elapsed = timer.time(step, args.steps, args.warmup)
    ids = pick_rows(nf, 96)
    post_err = check_rows(x, out[torch.from_numpy(ids).to(dev)].cpu().numpy(), ids, N_FFT, HOP, "mag")

    achieved_gbs = BYTES_PER_FRAME * nf / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = load_traffic()
    sus = sustained_leg(timer, step, kernel_ms, args.sustained_s)
    sus_gbs = BYTES_PER_FRAME * nf / (sus["ms_per_step"] * 1e-3) / 1e9
    ceil = copy_ceiling(aa, torch, dev, timer, xd.shape[0], nf * plan.cols)
    result = {
        "metric": "STFT frames/s (1024-pt, hop 256, 48 kHz f32)",
        "value": nf * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        # one GPU, one fixed job: there is nothing to scale; N > 1 keeps the TOTAL work fixed (configs[3], "strong"), and
        # the field says the same here so that the per-N lines of one sweep agree
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"configs[1]: batched 1024-pt STFT, hop 256, Hann, |X|, 10 min synthetic 48 kHz mono "
                        f"f32 ({nf} frames)",
            "n_fft": N_FFT, "hop": HOP, "output": "mag", "frames_total": nf, "kernel": plan.kernel,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_from_profile": traffic_src,
            "kernel_ms": kernel_ms, "bytes_per_frame": BYTES_PER_FRAME, "frames_per_launch": nf,
            # measured on this box, in this run (not a target: the nominal peak above stays `peak`)
            "copy_ceiling": {"copy_GBps": r4(ceil["copy_GBps"]), "mix_1r2w_GBps": r4(ceil["mix_1r2w_GBps"]),
                             "mix_1r4w_GBps": r4(ceil["mix_1r4w_GBps"]),
                             "frac_of_copy": r4(achieved_gbs / ceil["copy_GBps"]),
                             "frac_of_mix": r4(achieved_gbs / ceil["mix_1r2w_GBps"]), "copied_ok": ceil["copied_ok"],
                             "plain_alloc": ceil["plain_alloc"], "placement_probe_us": ceil["placement_probe_us"],
                             "how": ceil["how"]},
        },
        # >= 2 s of back-to-back launches of the same kernel between one pair of events
        "sustained_ms_per_step": sus["ms_per_step"], "sustained_frac": sus_gbs / HBM_PEAK_GBS,
        "sustained": {"launches": sus["launches"], "seconds": sus["seconds"],
                      "frames_per_s": nf / (sus["ms_per_step"] * 1e-3)},
        "post_check_max_err": r4(post_err), "post_check_rows": int(len(ids)), "post_check_tol": TOL_VS_TRUTH,
    }
    ok = post_err <= TOL_VS_TRUTH
    detail = dict(result)
    if not args.skip_configs:
        result["configs"] = other_configs(aa, torch, dev, timer, x, xd, args.config_steps, 20)
        detail["configs"] = result["configs"]
        for k, rec in result["configs"].items():
            for ek in ("err", "err_sine", "err_chirp"):
                if ek in rec and not rec[ek] <= 1e-5:
                    ok = False
                    print(f"post-check FAILED: configs[{k}].{ek} = {rec[ek]}", file=sys.stderr)
    if not args.skip_cpu:
        # PCIe-inclusive rates (never `value`): H2D samples + kernel + D2H rows
        host_out = np.empty((nf, plan.cols), dtype=np.float32)
        plan.execute_host(x[:N_FFT + 999 * HOP])
        t2 = time.perf_counter()
        plan.execute_host(x, out=host_out)
        detail["pcie_inclusive_frames_per_s"] = nf / (time.perf_counter() - t2)
        aa.pin(x)
        aa.pin(host_out)
        try:
            plan.execute_host(x, out=host_out)
            t3 = time.perf_counter()
            plan.execute_host(x, out=host_out)
            detail["pcie_inclusive_pinned_frames_per_s"] = nf / (time.perf_counter() - t3)
        finally:
            aa.unpin(host_out)
            aa.unpin(x)
        del host_out
        result["pcie_inclusive_pinned_frames_per_s"] = r4(detail["pcie_inclusive_pinned_frames_per_s"])
        result["cpu_baseline"] = cpu_baseline(x)
        detail["cpu_baseline"] = result["cpu_baseline"]
    write_detail(detail, 1)
    print(json.dumps(result), flush=True)
    if not ok:
        print(f"post-check FAILED: max-normalised error {post_err:.3e} > {TOL_VS_TRUTH}", file=sys.stderr)
        raise SystemExit(3)


def write_detail(detail, n):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"bench_detail_n{n}.json"), "w") as f:
            json.dump(detail, f, indent=1)
    except OSError:
        pass


# ---------------------------------------------------------------------------- N > 1 --
class HipCompute:
    """The per-rank compute of the sharded job: the fused HIP kernel behind the C ABI."""

    def __init__(self, n_fft, hop, dev):
        import torch

        import audio_analyzer_rs_amd as aa

        self.torch, self.dev = torch, dev
        self.plan = aa.Stft(n_fft, hop, output="mag")
        self.cols = self.plan.cols
        self.kernel = self.plan.kernel
        self.stream = torch.cuda.current_stream(dev)

    def to_device(self, x_np):
        return self.torch.from_numpy(x_np).to(self.dev)

    def empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.dev)

    def execute(self, xd, nf, out):
        if nf > 0:
            self.plan.execute_device(xd, 0, nf, out=out, stream=self.stream.cuda_stream)

    def execute_range(self, xd, fb, fe, rows):
        """frames [fb, fe) of the rank's range into their rows (one piece of the overlapped end-to-end leg)"""
        if fe > fb:
            self.plan.execute_device(xd, fb, fe, out=rows[fb:fe], stream=self.stream.cuda_stream)

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def events(self):
        return (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))


E2E_PIECES = 8        # pieces per rank of the overlapped end-to-end leg (scatter k + 1 || compute k || gather k - 1)


def run_sharded(dist, comp, world, rank, total_samples, steps, warmup, e2e_passes=3, ramp_s=0.25, ctx=None):
    """configs[3] on `world` ranks (one process group already initialised).  `comp` supplies the
    per-rank compute and its device memory (HipCompute here; the world-2 gloo test passes a CPU
    stand-in so that this very function runs in CI).  Returns rank 0's result dict, None elsewhere."""
    import torch

    import audio_analyzer_rs_amd as aa

    shard = aa.ShardPlan(total_samples, N_FFT, HOP, world, rank)
    nf = shard.local_frames
    x = aa.signals.synth_corpus(shard.s0, shard.local_samples)
    xd = comp.to_device(x)
    out = comp.empty(nf, comp.cols)

    def step():
        comp.execute(xd, nf, out)

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=out.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- (i) compute only: every rank transforms its own range, rows stay where they are --------
    t = time.perf_counter()
    while time.perf_counter() - t < ramp_s:
        for _ in range(20):
            step()
        comp.sync()
    for _ in range(warmup):
        step()
    comp.sync()
    dist.barrier()
    comp.sync()
    ev = comp.events()
    t0 = time.perf_counter()
    if ev:
        ev[0].record(comp.stream)
    for _ in range(steps):
        step()
    if ev:
        ev[1].record(comp.stream)
    comp.sync()
    elapsed = time.perf_counter() - t0         # this rank's K steps; the MAX over ranks is the job's
    dist.barrier()
    comp.sync()
    kernel_ms = ev[0].elapsed_time(ev[1]) / steps if ev else elapsed / steps * 1e3
    elapsed = max_over_ranks(elapsed)
    kernel_ms_max = max_over_ranks(kernel_ms)

    # The headline figure is complete here.  Rank 0 keeps the line it would print if nothing below came back (run_multi's
    # watchdog) and checks rows of its OWN shard against the truth - the end-to-end legs that follow are the first code
    # of this file that moves bytes between GPUs, and a fault there must not take the measured compute figure with it.
    total_frames = shard.total_frames
    achieved = BYTES_PER_FRAME * total_frames / (kernel_ms_max * 1e-3) / 1e9
    base = None
    if rank == 0:
        comp.execute(xd, nf, out)
        comp.sync()
        ids = pick_rows(nf, 64, seed=3)
        local_err = check_rows(x, out[torch.from_numpy(ids).to(out.device)].cpu().numpy(), ids, N_FFT, HOP, "mag")
        base = {
            "metric": "STFT frames/s (1024-pt, hop 256, 48 kHz f32)",
            "value": total_frames * steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "n_ranks_seen": dist.get_world_size(),
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[3]: {total_samples / SR / 60:.4g} min synthetic 48 kHz corpus "
                            f"({total_samples} samples, {total_frames} frames), 1024-pt STFT, hop 256, "
                            f"frame-range sharded over {world} ranks ({nf} frames on rank 0), |X| out",
                "n_fft": N_FFT, "hop": HOP, "output": "mag", "frames_total": total_frames,
                "parallelism": f"frame-range shards x{world}; value = compute only, rows stay sharded, "
                               "no data-path collective",
                "kernel": comp.kernel,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                "kernel_ms": kernel_ms_max, "bytes_per_frame": BYTES_PER_FRAME,
                "frames_per_launch": nf,
            },
            "post_check_local_max_err": local_err,
            "post_check_max_err": local_err,
            "post_check_tol": TOL_VS_TRUTH,
        }
        if ctx is not None:
            ctx["partial"] = dict(base, e2e={"error": "the end-to-end legs did not come back (watchdog)"})
    try:
        e2e_rec, post_err = _e2e_legs(dist, comp, aa, torch, shard, xd, out, nf, rank, e2e_passes, max_over_ranks, ctx)
    except Exception as ex:                                       # noqa: BLE001 (reported in the line, never fatal for it)
        e2e_rec, post_err = {"error": f"{type(ex).__name__}: {ex}"}, None
    if rank != 0:
        return None
    base["e2e"] = e2e_rec
    if post_err is not None:
        base["post_check_max_err"] = max(post_err, base["post_check_local_max_err"])
    return base


def _e2e_legs(dist, comp, aa, torch, shard, xd, out, nf, rank, e2e_passes, max_over_ranks, ctx):
    """configs[3] end to end (rank 0 owns the samples and wants all the rows): serial and overlapped.  Returns the `e2e`
    record and the post-check error of the gathered rows (None off rank 0)."""
    # ---- (ii) end to end: rank 0 owns the samples and wants all the rows -------------------------
    full = aa.gather_samples(xd, shard, dst=0)                     # untimed: builds rank 0's input
    rows_root = comp.empty(shard.total_frames, comp.cols) if rank == 0 else None
    phases = []
    for p in range(e2e_passes + 1):                                # pass 0 is the warm-up
        dist.barrier()
        comp.sync()
        ta = time.perf_counter()
        mine = aa.scatter_samples(full, shard, device=out.device, src=0)
        comp.sync()
        tb = time.perf_counter()
        comp.execute(mine, nf, out)
        comp.sync()
        tc = time.perf_counter()
        aa.gather_rows(out, shard, dst=0, out=rows_root)
        comp.sync()
        td = time.perf_counter()
        if p:
            phases.append([max_over_ranks(tb - ta), max_over_ranks(tc - tb), max_over_ranks(td - tc),
                           max_over_ranks(td - ta)])
    e2e = np.median(np.array(phases), axis=0)

    # ---- post-check on rank 0: gathered rows (every rank's block) against the truth --------------
    def post_check(slab):
        ids = pick_rows(shard.total_frames, 64)
        for f0, f1 in shard.all_frame_ranges():                    # both edges of every shard
            ids = np.union1d(ids, [f0, max(f0, f1 - 1)])
        for r, (f0, f1) in enumerate(shard.all_frame_ranges()):    # and both edges of every piece of the overlapped leg
            for k in range(E2E_PIECES):
                p0, p1 = aa.shard_piece(f0, f1, E2E_PIECES, k)
                ids = np.union1d(ids, [p0, max(p0, p1 - 1)])
        ids = ids[ids < shard.total_frames].astype(np.int64)
        return check_rows(full.cpu().numpy(), slab[torch.from_numpy(ids).to(slab.device)].cpu().numpy(), ids, N_FFT, HOP, "mag")

    post_err = post_check(rows_root) if rank == 0 else None

    # ---- (iii) the same job with the three legs overlapped: every rank's range in E2E_PIECES pieces, piece k + 1's
    # samples on their way while piece k is transformed and piece k - 1's rows travel back (aa.overlapped_e2e) ----
    if rank == 0:
        rows_root.fill_(float("nan"))
    over = []
    for p in range(e2e_passes + 1):
        dist.barrier()
        comp.sync()
        ta = time.perf_counter()
        trace = []
        # (off the root the rank's buffers live across the passes: a transport that maps destinations into the sender -
        # the one-GPU rehearsal's - maps them once)
        aa.overlapped_e2e(full, shard, comp.execute_range, comp.cols, n_pieces=E2E_PIECES, src=0, device=out.device,
                          out=rows_root, trace=trace, local=None if rank == 0 else xd, rows=None if rank == 0 else out)
        comp.sync()
        tb = time.perf_counter()
        if p:
            over.append(max_over_ranks(tb - ta))
    overlapped = float(np.median(over))
    # the schedule itself, counted (not timed): computes queued with the next scatter / an earlier gather in flight
    sched = aa.overlap_counts(trace, E2E_PIECES)
    post_err_overlapped = post_check(rows_root) if rank == 0 else None
    if ctx is not None:
        ctx.update(shard=shard, full=full, out=out, rows_root=rows_root, nf=nf)
    if rank != 0:
        return None, None
    total_frames, world = shard.total_frames, dist.get_world_size()
    rec = {
        "what": "rank 0 scatters sample ranges (grouped send/recv), every rank computes, rows "
                "gathered unpadded to rank 0; median of %d passes, max over ranks" % e2e_passes,
        "scatter_ms": r4(e2e[0] * 1e3), "compute_ms": r4(e2e[1] * 1e3), "gather_ms": r4(e2e[2] * 1e3),
        "serial_total_ms": r4(e2e[3] * 1e3), "serial_frames_per_s": r4(total_frames / e2e[3]),
        "gather_GBps": r4(total_frames * comp.cols * 4 * (world - 1) / world / e2e[2] / 1e9),
        # the overlapped form of the same job: scatter k + 1 || compute k || gather k - 1
        "total_ms": r4(overlapped * 1e3), "frames_per_s": r4(total_frames / overlapped),
        "overlap": {"pieces_per_rank": E2E_PIECES, "total_ms": r4(overlapped * 1e3),
                    "speedup_vs_serial": r4(e2e[3] / overlapped),
                    "post_check_max_err": post_err_overlapped,
                    # counted, not timed: computes queued with the next piece's scatter / an earlier piece's gather
                    # posted and not yet waited for (pieces - 1 each when the legs overlap as designed)
                    "computes_with_scatter_in_flight": sched["scatter_ahead"],
                    "computes_with_gather_in_flight": sched["gather_behind"],
                    "how": "aa.overlapped_e2e: batch_isend_irecv per piece, waited for where the data is needed"},
    }
    return rec, (post_err if post_err_overlapped is None else max(post_err, post_err_overlapped))


def run_multi(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # AA_BENCH_REHEARSAL=1 (development only): several ranks share the GPUs of a smaller box and
    # rendezvous over gloo, to exercise the N > 1 code path where no N-GPU node is available.
    # AA_BENCH_REHEARSAL=ipc (r5): the same, but the ranks that share a GPU exchange THROUGH DEVICE MEMORY (torch IPC
    # handles, interprocess events: aa.IpcTransport) - the schedule of the overlapped leg on device buffers, no host staging.
    mode = os.environ.get("AA_BENCH_REHEARSAL", "0")
    rehearsal = mode in ("1", "ipc")
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    transport = None
    if mode == "ipc":
        import audio_analyzer_rs_amd as aa

        if torch.cuda.device_count() != 1:
            raise SystemExit("AA_BENCH_REHEARSAL=ipc is the ONE-GPU rehearsal (every rank on device 0)")
        dist.init_process_group("gloo")
        comp = HipCompute(N_FFT, HOP, dev)
        transport = aa.IpcTransport(dist, dev)
        aa.set_transport(transport.post)
        hip_sync = comp.sync
        comp.sync = lambda: (hip_sync(), transport.drain())
    elif rehearsal:
        dist.init_process_group("gloo")
        comp = RehearsalCompute(N_FFT, HOP, dev)
    else:
        dist.init_process_group("nccl", device_id=dev)
        comp = HipCompute(N_FFT, HOP, dev)
    total = int(os.environ.get("AA_BENCH_CORPUS_SAMPLES", SAMPLES_1H))
    ctx = {}
    rc = 0
    # run_sharded under a watchdog: its end-to-end legs are the first code that moves bytes between GPUs; if they never
    # come back, rank 0 still prints the compute-only line it has held since the timed steps (ctx["partial"])
    import threading

    box = {}

    def body():
        try:
            torch.cuda.set_device(dev_index)                 # (the current device is per thread)
            box["result"] = run_sharded(dist, comp, world, rank, total, args.steps, args.warmup, ctx=ctx)
        except BaseException as ex:                          # noqa: BLE001 (re-raised below, on the main thread)
            box["exc"] = ex

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(float(os.environ.get("AA_BENCH_E2E_TIMEOUT", "420")))
    hung = th.is_alive()
    if "exc" in box:
        raise box["exc"]
    if hung:
        print(f"rank {rank}: the sharded run did not come back within its limit", file=sys.stderr, flush=True)
        if rank != 0 or "partial" not in ctx:
            os._exit(4 if rank == 0 else 0)
    result = ctx["partial"] if hung else box.get("result")
    if transport is not None and not hung:
        transport.close()
    if rank == 0:
        if mode == "ipc":
            result["config"]["parallelism"] += (" (REHEARSAL: the ranks share ONE GPU and exchange through its memory - IPC "
                                                "handles, interprocess events; no link is crossed)")
        elif rehearsal:
            result["config"]["parallelism"] += " (REHEARSAL: ranks share GPUs, gloo, host-staged exchanges)"
        write_detail(result, world)
        print(json.dumps(result), flush=True)            # the ONE line; nothing below can take it back
        if not result["post_check_max_err"] <= TOL_VS_TRUTH:
            print(f"post-check FAILED: {result['post_check_max_err']}", file=sys.stderr)
            rc = 3
        if "error" in result.get("e2e", {}):
            print("e2e legs: " + result["e2e"]["error"], file=sys.stderr, flush=True)
    if hung:
        os._exit(rc)                                       # the line is out; the legs that hang hold the process group
    if not rehearsal and "shard" in ctx and os.environ.get("AA_BENCH_CABI_LEG", "1") == "1":
        # The same end-to-end pass on the C-ABI collectives (aa_comm_*: librccl bound by libaa_stft.so,
        # what a Rust / C++ host would call).  Reported on stderr and in the detail file; it runs in
        # a watchdog thread so that a communicator that never forms cannot hold the job.
        import threading

        box = {}
        th = threading.Thread(target=cabi_leg, args=(dist, comp, world, rank, total, ctx, box), daemon=True)
        th.start()
        th.join(float(os.environ.get("AA_BENCH_CABI_TIMEOUT", "120")))
        if rank == 0:
            rec = box.get("result") or {"error": box.get("error", "timeout")}
            print("e2e_cabi " + json.dumps(rec), file=sys.stderr, flush=True)
            result["e2e_cabi"] = rec
            write_detail(result, world)
        if th.is_alive():
            os._exit(rc)                                   # the line is out; do not wait on a hung RCCL call
    try:
        dist.destroy_process_group()
    except Exception:
        pass
    if rc:
        raise SystemExit(rc)


def cabi_leg(dist, comp, world, rank, total_samples, ctx, box, passes=3):
    """scatter + compute + gather on aa_scatter_samples / aa_gather_rows (include/aa_stft.h)."""
    try:
        import torch

        import audio_analyzer_rs_amd as aa

        torch.cuda.set_device(comp.dev)
        ok = torch.tensor([1.0 if "rccl" in aa.Comm.backend() and "not found" not in aa.Comm.backend() else 0.0],
                          device=comp.dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)            # every rank could bind librccl, or nobody goes on
        if float(ok.item()) < 1.0:
            box["error"] = "librccl not bound on every rank: " + aa.Comm.backend()
            return
        uid = [aa.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = aa.Comm(uid[0], world, rank)
        shard, full, out, rows_root, nf = (ctx[k] for k in ("shard", "full", "out", "rows_root", "nf"))
        st = comp.stream.cuda_stream
        times = []
        for p in range(passes + 1):
            comm.barrier(st)
            ta = time.perf_counter()
            mine = comm.scatter_samples(full, total_samples, N_FFT, HOP, root=0, stream=st)
            comp.sync()
            tb = time.perf_counter()
            comp.execute(mine, nf, out)
            comp.sync()
            tc = time.perf_counter()
            comm.gather_rows(out, shard.total_frames, root=0, out=rows_root, stream=st)
            comp.sync()
            td = time.perf_counter()
            if p:
                times.append([tb - ta, tc - tb, td - tc, td - ta])
        comm.barrier(st)
        t = torch.tensor(np.median(np.array(times), axis=0), dtype=torch.float64, device=comp.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e = t.cpu().numpy()

        def post_check():
            ids = pick_rows(shard.total_frames, 64, seed=5)
            for f0, f1 in shard.all_frame_ranges():
                ids = np.union1d(ids, [f0, max(f0, f1 - 1)])
                for k in range(E2E_PIECES):
                    p0, p1 = aa.shard_piece(f0, f1, E2E_PIECES, k)
                    ids = np.union1d(ids, [p0, max(p0, p1 - 1)])
            ids = ids[ids < shard.total_frames].astype(np.int64)
            return check_rows(full.cpu().numpy(), rows_root[torch.from_numpy(ids).to(comp.dev)].cpu().numpy(),
                              ids, N_FFT, HOP, "mag")

        err = post_check() if rank == 0 else None
        # the overlapped form: aa_stft_execute_sharded, gathers on a communicator of their own
        uid2 = [aa.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid2, src=0)
        comm_out = aa.Comm(uid2[0], world, rank)
        if rank == 0:
            rows_root.fill_(float("nan"))
        over = []
        for p in range(passes + 1):
            comm.barrier(st)
            ta = time.perf_counter()
            comm.execute_sharded(comp.plan, full, total_samples, root=0, n_pieces=E2E_PIECES, comm_out=comm_out,
                                 out=rows_root, stream=st)
            comp.sync()
            tb = time.perf_counter()
            if p:
                over.append(tb - ta)
        comm.barrier(st)
        t = torch.tensor([float(np.median(over))], dtype=torch.float64, device=comp.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ov = float(t.item())
        err_over = post_check() if rank == 0 else None
        comm_out.close()
        comm.close()
        box["result"] = {"backend": aa.Comm.backend(), "scatter_ms": r4(e[0] * 1e3), "compute_ms": r4(e[1] * 1e3),
                         "gather_ms": r4(e[2] * 1e3), "serial_total_ms": r4(e[3] * 1e3),
                         "total_ms": r4(ov * 1e3), "frames_per_s": r4(shard.total_frames / ov),
                         "overlap": {"pieces_per_rank": E2E_PIECES, "speedup_vs_serial": r4(e[3] / ov),
                                     "post_check_max_err": err_over,
                                     "how": "aa_stft_execute_sharded: three streams, scatters and gathers on two communicators"},
                         "post_check_max_err": err}
    except Exception as ex:                                  # noqa: BLE001 (reported, never fatal)
        box["error"] = f"{type(ex).__name__}: {ex}"


class RehearsalCompute(HipCompute):
    """gloo cannot move device memory: the exchanges run on host tensors, the compute on the GPU."""

    def to_device(self, x_np):
        self._xd = self.torch.from_numpy(x_np).to(self.dev)
        return self.torch.from_numpy(x_np)

    def empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float32)

    def execute(self, x_host, nf, out):
        if nf > 0:
            xd = x_host.to(self.dev)
            out.copy_(self.plan.execute_device(xd, 0, nf, stream=self.stream.cuda_stream).cpu())

    def execute_range(self, x_host, fb, fe, rows):
        if fe > fb:
            xd = x_host[fb * self.plan.hop:(fe - 1) * self.plan.hop + self.plan.n_fft].to(self.dev)
            rows[fb:fe].copy_(self.plan.execute_device(xd, 0, fe - fb, stream=self.stream.cuda_stream).cpu())

    def events(self):
        return None


def spawn_ranks(args):
    """WORLD_SIZE unset and --gpus N > 1: start the N ranks as fresh child processes.  Nothing in
    this process has touched the GPU (no exec of a GPU-initialised process, see the image notes)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config-steps", type=int, default=200, help="timed launches per entry of `configs`")
    ap.add_argument("--skip-cpu", action="store_true", help="omit the cpu_baseline / PCIe legs (profiling runs)")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="length of the sustained leg (counter passes shorten it)")
    ap.add_argument("--skip-configs", action="store_true", help="headline kernel only (profiling runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    # AA_BENCH_FORCE_SHARDED=1 with --gpus 1 under torch.distributed.run: the N > 1 code path (RCCL
    # process group, barrier, MAX all-reduce, scatter / gather legs, aa_comm_*) at world size 1 — the
    # rehearsal a one-GPU box allows.
    if args.gpus > 1 or (os.environ.get("AA_BENCH_FORCE_SHARDED") == "1" and "WORLD_SIZE" in os.environ):
        run_multi(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()

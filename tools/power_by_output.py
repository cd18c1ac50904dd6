"""Package power and sclk (rocm-smi, 5 samples a second) while one plan runs back to back for a few
seconds per output kind: what the chip draws under the magnitude, complex, log-mel and MFCC variants of
the 1024 / 256 kernel and under the 2048 / 512 and 256 / 64 magnitude kernels.  Run on a GPU box:
    python tools/power_by_output.py [seconds per variant]          (diagnostic, not shipped)"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sample(stop, out):
    while not stop.is_set():
        t = time.time()
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            txt = ""
        p = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
        if p and c:
            out.append((t, float(p.group(1)), int(c.group(1))))
        time.sleep(0.15)


def main():
    import torch
    import audio_analyzer_rs_amd as aa

    sec = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    x = aa.signals.synth_corpus(0, 28_800_000)
    xd = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream()
    samples, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(stop, samples), daemon=True)
    time.sleep(1.0)
    th.start()
    time.sleep(1.5)
    idle = [p for _, p, _ in samples]
    print(f"# idle: {sum(idle) / max(len(idle), 1):.0f} W")
    for n, hop, kind, bpf in ((1024, 256, "mag", 3076), (1024, 256, "complex", 5128), (1024, 256, "logmel", 1536),
                              (1024, 256, "mfcc", 1184), (2048, 512, "mag", 6148), (256, 64, "mag", 772), (1024, 256, "mag", 3076)):
        kw = dict(n_mels=128) if kind in ("logmel", "mfcc") else {}
        if kind == "mfcc":
            kw["n_mfcc"] = 40
        plan = aa.Stft(n, hop, output=kind, **kw)
        nf = plan.num_frames(xd.shape[0])
        out = torch.empty((nf, plan.cols), dtype=torch.float32, device="cuda")
        t0 = time.time()
        last = 0.0
        while time.time() - t0 < sec:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(500):
                plan.execute_device(xd, 0, nf, out=out, stream=st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            last = e0.elapsed_time(e1) / 500 * 1e3
        t1 = time.time()
        win = [(p, c) for t, p, c in samples if t0 + 1.5 < t < t1]
        if win:
            pw, ck = sum(p for p, _ in win) / len(win), sum(c for _, c in win) / len(win)
            print(f"{n:5d}/{hop:<4d} {kind:8s} {last:7.2f} us/launch  {bpf * nf / last / 1e6:6.3f} TB/s algorithmic  {pw:7.1f} W  "
                  f"sclk {ck:5.0f} MHz  {pw * last * 1e-3:6.1f} mJ/launch ({len(win)} samples)")
        else:
            print(f"{n:5d}/{hop:<4d} {kind:8s} {last:7.2f} us/launch  no power samples")
        time.sleep(1.0)
    stop.set()


if __name__ == "__main__":
    main()

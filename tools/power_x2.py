"""Package power and sclk (rocm-smi, ~5 samples a second) under the two 1024 / 256 magnitude kernels, each back to
back for a few seconds, alternating: AA_W1024_X2=0 (8 x 8 x 8) and 1 (two frames per wave).  Run on a GPU box:
    python tools/power_x2.py [seconds per leg] [rounds]          (diagnostic, not shipped)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.power_by_output import sample


def main():
    import torch
    import audio_analyzer_rs_amd as aa
    if os.environ.get("AA_TOOL_LIB"):
        aa._lib.LIB_PATH = os.path.abspath(os.environ["AA_TOOL_LIB"])
    sec = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    xd = torch.from_numpy(aa.signals.synth_corpus(0, 28_800_000)).cuda()
    st = torch.cuda.current_stream()
    plans = {}
    for v in ("0", "1"):
        os.environ["AA_W1024_X2"] = v
        plans[v] = aa.Stft(1024, 256, output="mag")
    nf = plans["0"].num_frames(xd.shape[0])
    out = torch.empty((nf, 513), dtype=torch.float32, device="cuda")
    samples, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(stop, samples), daemon=True)
    th.start()
    time.sleep(2.0)
    idle = [p for _, p, _ in samples]
    print(f"# idle: {sum(idle) / max(len(idle), 1):.0f} W")
    for r in range(rounds):
        for v in ("0", "1"):
            plan = plans[v]
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
            pw = sum(p for p, _ in win) / max(len(win), 1)
            ck = sum(c for _, c in win) / max(len(win), 1)
            print(f"{plan.kernel:11s} {last:7.2f} us/launch  {pw:7.1f} W  sclk {ck:5.0f} MHz  {pw * last * 1e-3:6.1f} mJ/launch "
                  f"({len(win)} samples)", flush=True)
            time.sleep(1.0)
    stop.set()


if __name__ == "__main__":
    main()

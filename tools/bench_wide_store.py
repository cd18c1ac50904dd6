"""Times the 1024/256 magnitude kernel with the output slab at a 16-byte-aligned and at a
4-byte-aligned base (the WIDE / unaligned store paths).  Run by hand on a GPU box:
    python tools/bench_wide_store.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import audio_analyzer_rs_amd as aa
    x = aa.signals.synth_corpus(0, 28_800_000)
    xd = torch.from_numpy(x).cuda()
    plan = aa.Stft(1024, 256, output="mag")
    nf = plan.num_frames(xd.shape[0])
    buf = torch.empty(nf * 513 + 16, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for off in (0, 1, 0, 1):
        out = buf[off:off + nf * 513].view(nf, 513)
        for _ in range(300):
            plan.execute_device(xd, 0, nf, out=out, stream=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            plan.execute_device(xd, 0, nf, out=out, stream=st)
        e1.record()
        torch.cuda.synchronize()
        print("offset", off, "floats:", e0.elapsed_time(e1) / 200 * 1e3, "us")


if __name__ == "__main__":
    main()

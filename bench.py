"""Headline benchmark: batched 1024-pt STFT (hop 256, periodic Hann, |X|) over synthetic
48 kHz mono f32 — BASELINE.json configs[1] ("1xMI355X: batched 1024-pt STFT over 10 min").

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (frame + window + real FFT + magnitude, one fused
kernel) over one rank's 10-minute buffer, input and output resident in HBM.  With N > 1
(launched by torch.distributed.run, one rank per GPU) the corpus is N x 10 min, frame-
range sharded with the n_fft-hop halo; every rank synthesises its own samples, computes
its own rows and keeps them (no data-path collective) -> weak scaling.

Rank 0 prints ONE JSON line; value = total frames of all ranks per second of the slowest
rank.  `roofline` prices the fused kernel against HBM with the ALGORITHMIC bytes per frame
(hop*4 in + (n_fft/2+1)*4 out = 3076 B); `cpu_baseline` times the CPU
restatement (oracle/, a port of the reference's path — the Rust crate cannot be built
here) on the same workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FFT, HOP, SR = 1024, 256, 48000
SAMPLES_PER_RANK = 600 * SR                  # 10 min
BYTES_PER_FRAME = HOP * 4 + (N_FFT // 2 + 1) * 4   # 3076: algorithmic bytes
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(x, min_seconds=10.0, max_reps=20):
    """Time the CPU restatement (oracle) on rank 0's full 10-min buffer, single thread,
    repeated until >= min_seconds of CPU work; then once on all host cores."""
    import oracle

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

    # all cores: frames partitioned across threads (ctypes releases the GIL)
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
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": single,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"the full 10-min buffer ({nf} frames) x {reps} passes, {dt:.1f} s, 1 thread; "
                  "C restatement of the reference path, gcc -O3 -mavx2 -ffp-contract=off",
        "all_cores": {"value": multi, "cores": cores, "passes": mreps, "seconds": round(mdt, 2)},
        "cpu_model": cpu_model,
    }


def load_traffic():
    """HBM bytes per launch from the committed PMC passes (profiles/*_pmc_traffic.json)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--skip-cpu", action="store_true", help="omit the cpu_baseline leg (profiling runs)")
    ap.add_argument("--output", default="mag", choices=["mag", "complex", "power", "logpower"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import audio_analyzer_rs_amd as aa

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # AA_BENCH_REHEARSAL=1 (development only): several ranks share the GPUs of a smaller box and
    # rendezvous over gloo, to exercise the N > 1 code path where no N-GPU node is available.
    rehearsal = os.environ.get("AA_BENCH_REHEARSAL", "0") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- this rank's shard of the N x 10 min corpus -------------------------------
    total_samples = world * SAMPLES_PER_RANK
    shard = aa.ShardPlan(total_samples, N_FFT, HOP, world, rank)
    x = aa.signals.synth_corpus(shard.s0, shard.local_samples)
    xd = torch.from_numpy(x).to(dev)
    plan = aa.Stft(N_FFT, HOP, output=args.output)
    assert plan.kernel == "wave1024", plan.kernel
    nf = shard.local_frames
    assert plan.num_frames(xd.shape[0]) == nf
    out = torch.empty((nf, plan.cols), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        plan.execute_device(xd, 0, nf, out=out, stream=stream.cuda_stream)

    # Device bring-up (not a step): a cold MI355X takes ~30 ms of back-to-back launches to
    # reach its steady clock / page-table state (first 300 launches measured 90 us each, every
    # later batch 80 us - tools/launch_overhead.py).  Then the W untimed warm-up steps.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.25:
        for _ in range(50):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    # this rank's K steps are done here; the closing barrier's own latency is not a step, the
    # MAX over ranks below is what makes the figure the whole job's
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_frames = shard.total_frames
    value = total_frames * args.steps / elapsed
    achieved_gbs = BYTES_PER_FRAME * nf / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": "STFT frames/s (1024-pt, hop 256, 48 kHz f32)",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: batched 1024-pt STFT, hop 256, periodic Hann, |X| out, "
                        f"10 min synthetic 48 kHz mono f32 per GPU ({nf} frames/GPU)",
            "n_fft": N_FFT, "hop": HOP, "output": args.output,
            "frames_total": total_frames, "samples_per_gpu": int(shard.local_samples),
            "parallelism": f"frame-range shards x{world}, outputs stay sharded" + (" (REHEARSAL: ranks share GPUs, gloo)" if rehearsal else ""),
            "kernel": plan.kernel,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": load_traffic(),
            "kernel_ms": kernel_ms,
            "bytes_per_frame": BYTES_PER_FRAME,
            "frames_per_launch": nf,
        },
    }
    if world == 1 and not args.skip_cpu:
        # PCIe-inclusive rate (never `value`): H2D samples + kernel + D2H rows, host buffers.
        host_out = np.empty((nf, plan.cols), dtype=np.float32)
        plan.execute_host(x[:N_FFT + 999 * HOP])
        t2 = time.perf_counter()
        plan.execute_host(x, out=host_out)
        result["pcie_inclusive_frames_per_s"] = nf / (time.perf_counter() - t2)
        # same with page-locked caller buffers (aa_host_register): chunked two-stream pipeline
        aa.pin(x)
        aa.pin(host_out)
        try:
            plan.execute_host(x, out=host_out)
            t3 = time.perf_counter()
            plan.execute_host(x, out=host_out)
            result["pcie_inclusive_pinned_frames_per_s"] = nf / (time.perf_counter() - t3)
        finally:
            aa.unpin(host_out)
            aa.unpin(x)
        del host_out
        result["cpu_baseline"] = cpu_baseline(x)
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

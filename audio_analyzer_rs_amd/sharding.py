"""Frame-range sharding of one STFT job over the GPUs of a node.

Frames are independent units (each reads n_fft consecutive samples and writes one row),
so rank r of R owns frames [floor(r*F/R), floor((r+1)*F/R)) and needs the samples
[f0*hop, (f1-1)*hop + n_fft) — a halo of n_fft-hop samples shared with the next rank.
Frame indices stay global, so frame -> sample mapping is bit-exact for any world size.

There is no data-path collective: each rank computes its rows and keeps them.  Two
optional exchanges over torch.distributed (backend "nccl" == RCCL over xGMI on the GPU
box, "gloo" in the CPU tests) are provided for callers that need them:
  scatter_samples  rank 0 holds the whole buffer -> every rank receives its range
  gather_rows      every rank's rows -> one [F, cols] tensor on dst (or on all ranks)
"""
import ctypes

from . import _lib


def shard_frames(total_frames, world, rank):
    lib = _lib.load()
    f0, f1 = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.aa_shard_frames(int(total_frames), int(world), int(rank), ctypes.byref(f0),
                                   ctypes.byref(f1)))
    return f0.value, f1.value


def shard_samples(f0, f1, n_fft, hop):
    lib = _lib.load()
    s0, s1 = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.aa_shard_samples(int(f0), int(f1), int(n_fft), int(hop), ctypes.byref(s0),
                                    ctypes.byref(s1)))
    return s0.value, s1.value


def num_frames(n_samples, n_fft, hop):
    return 0 if n_samples < n_fft else (n_samples - n_fft) // hop + 1


class ShardPlan:
    """Where rank `rank` of `world` sits in a job of `n_samples` samples."""

    def __init__(self, n_samples, n_fft, hop, world, rank):
        self.n_samples, self.n_fft, self.hop = int(n_samples), int(n_fft), int(hop)
        self.world, self.rank = int(world), int(rank)
        self.total_frames = num_frames(self.n_samples, self.n_fft, self.hop)
        self.f0, self.f1 = shard_frames(self.total_frames, self.world, self.rank)
        self.s0, self.s1 = shard_samples(self.f0, self.f1, self.n_fft, self.hop)

    @property
    def local_frames(self):
        return self.f1 - self.f0

    @property
    def local_samples(self):
        return self.s1 - self.s0

    def all_frame_ranges(self):
        return [shard_frames(self.total_frames, self.world, r) for r in range(self.world)]


def scatter_samples(full, shard, device=None, src=0):
    """rank `src` holds `full` (1-D float32 tensor); every rank returns its [s0, s1) slice.
    Point-to-point sends (a broadcast would move world x more bytes than needed)."""
    import torch
    import torch.distributed as dist

    dev = device if device is not None else (full.device if full is not None else "cpu")
    local = torch.empty(shard.local_samples, dtype=torch.float32, device=dev)
    if shard.world == 1:
        local.copy_(full[shard.s0:shard.s1])
        return local
    if shard.rank == src:
        reqs = []
        for r in range(shard.world):
            f0, f1 = shard_frames(shard.total_frames, shard.world, r)
            s0, s1 = shard_samples(f0, f1, shard.n_fft, shard.hop)
            if r == src:
                local.copy_(full[s0:s1])
            elif s1 > s0:
                reqs.append(dist.isend(full[s0:s1].contiguous(), dst=r))
        for q in reqs:
            q.wait()
    elif shard.local_samples > 0:
        dist.recv(local, src=src)
    return local


def gather_rows(rows, shard, dst=None):
    """rows: this rank's [local_frames, cols] tensor.  dst=None -> all ranks get [F, cols]
    (all_gather on padded blocks); dst=r -> only rank r does (others return None)."""
    import torch
    import torch.distributed as dist

    if shard.world == 1:
        return rows
    cols = rows.shape[1]
    ranges = shard.all_frame_ranges()
    cap = max(f1 - f0 for f0, f1 in ranges)
    pad = torch.zeros((cap, cols), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    if dst is None:
        blocks = [torch.empty_like(pad) for _ in range(shard.world)]
        dist.all_gather(blocks, pad)
    else:
        blocks = [torch.empty_like(pad) for _ in range(shard.world)] if shard.rank == dst else None
        dist.gather(pad, blocks, dst=dst)
        if shard.rank != dst:
            return None
    return torch.cat([b[:f1 - f0] for b, (f0, f1) in zip(blocks, ranges)], dim=0)

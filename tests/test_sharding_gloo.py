"""CPU tests of the N > 1 path: frame-range sharding with the n_fft-hop halo, the optional
scatter / gather exchanges over torch.distributed (gloo here, RCCL on the GPU box), and
bit-exact global frame indexing.  The per-rank compute is the CPU oracle (a stand-in for
the HIP path, which needs a GPU) — what is under test is the partition and the plumbing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_samples, n_fft, hop, tmpdir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import audio_analyzer_rs_amd as aa
        import oracle

        shard = aa.ShardPlan(n_samples, n_fft, hop, world, rank)
        # (a) every rank synthesises its own range: identical to slicing the whole corpus
        whole = aa.signals.synth_corpus(0, n_samples)
        mine = aa.signals.synth_corpus(shard.s0, shard.local_samples)
        assert np.array_equal(mine, whole[shard.s0:shard.s1])
        # (b) rank 0 scatters ranges point-to-point
        full_t = torch.from_numpy(whole) if rank == 0 else None
        got = aa.scatter_samples(full_t, shard, device="cpu", src=0)
        assert np.array_equal(got.numpy(), whole[shard.s0:shard.s1])
        # per-rank compute on LOCAL frame indices; rows land at global rows f0..f1
        assert oracle.num_frames(shard.local_samples, n_fft, hop) == shard.local_frames
        rows = oracle.stft(mine, n_fft, hop, kind="mag")
        rows_t = torch.from_numpy(rows)
        # (c) all-gather and gather-to-root reproduce the single-process result bit for bit
        ref = oracle.stft(whole, n_fft, hop, kind="mag")
        allr = aa.gather_rows(rows_t, shard, dst=None)
        assert allr.shape == ref.shape and np.array_equal(allr.numpy(), ref)
        root = aa.gather_rows(rows_t, shard, dst=world - 1)
        if rank == world - 1:
            assert np.array_equal(root.numpy(), ref)
        else:
            assert root is None
        # frame -> sample mapping is global: local frame i of this rank is global frame f0+i
        for i in (0, shard.local_frames - 1):
            g = shard.f0 + i
            assert np.array_equal(mine[i * hop:i * hop + n_fft], whole[g * hop:g * hop + n_fft])
        # (d) the overlapped form - scatter k + 1 || compute k || gather k - 1 over pieces of every rank's range -
        # yields the same slab bit for bit, for piece counts below, at and above the frames a rank owns; the
        # pieces each rank transforms tile its range in order
        for n_pieces, src in ((1, 0), (3, 0), (8, world - 1), (shard.local_frames + 5, 0)):
            calls = []

            def execute(local, fb, fe, rows_out):
                calls.append((fb, fe))
                part = oracle.stft(local.numpy()[fb * hop:(fe - 1) * hop + n_fft], n_fft, hop, kind="mag")
                rows_out[fb:fe].copy_(torch.from_numpy(part))

            full_src = torch.from_numpy(whole.copy()) if rank == src else None
            loc, rws, slab = aa.overlapped_e2e(full_src, shard, execute, ref.shape[1], n_pieces=n_pieces, src=src)
            assert np.array_equal(loc.numpy(), whole[shard.s0:shard.s1])
            assert np.array_equal(rws.numpy(), ref[shard.f0:shard.f1])
            assert (slab is not None) == (rank == src)
            if rank == src:
                assert np.array_equal(slab.numpy(), ref)
            want = [aa.shard_piece(shard.f0, shard.f1, n_pieces, k) for k in range(n_pieces)]
            assert calls == [(a - shard.f0, b - shard.f0) for a, b in want if b > a]
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_samples,n_fft,hop", [(2, 48_000, 1024, 256), (3, 20_011, 256, 64),
                                                      (2, 1024 + 256, 1024, 256), (8, 60_000, 1024, 256)])
def test_sharded_job_matches_single_process(tmp_path, world, n_samples, n_fft, hop):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_samples, n_fft, hop, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"ok{r}").exists()


def test_shard_plan_properties():
    import sys

    sys.path.insert(0, ROOT)
    import audio_analyzer_rs_amd as aa

    # BASELINE config 4: 1 h corpus over 8 GPUs, 84 375 +- 1 frames per rank
    plans = [aa.ShardPlan(172_800_000, 1024, 256, 8, r) for r in range(8)]
    assert plans[0].total_frames == 674_997
    assert sum(p.local_frames for p in plans) == 674_997
    assert all(abs(p.local_frames - 84_375) <= 1 for p in plans)
    for a, b in zip(plans, plans[1:]):
        assert a.f1 == b.f0
        assert a.s1 - b.s0 == 1024 - 256            # the halo
    # more ranks than frames: empty shards are well formed
    tiny = [aa.ShardPlan(1024 + 256, 1024, 256, 4, r) for r in range(4)]
    assert [p.local_frames for p in tiny] == [0, 1, 0, 1]
    assert all(p.local_samples == (1024 if p.local_frames else 0) for p in tiny)


def test_pieces_tile_frames_and_samples():
    """aa_shard_piece / piece_samples: the pieces of a frame range tile it, and the samples they add tile the
    range's samples exactly (every sample travels once, consecutive pieces share the n_fft - hop halo)."""
    import sys

    sys.path.insert(0, ROOT)
    import audio_analyzer_rs_amd as aa

    for n_fft, hop in ((1024, 256), (2048, 512), (256, 64), (1024, 1000)):
        for f0, f1 in ((0, 0), (5, 6), (100, 117), (0, 1000), (84_375, 168_750)):
            s0, s1 = aa.shard_samples(f0, f1, n_fft, hop)
            for P in (1, 2, 3, 8, 16, 64):
                edge_f, edge_s = f0, s0
                for k in range(P):
                    p0, p1 = aa.shard_piece(f0, f1, P, k)
                    a, b = aa.piece_samples(f0, f1, P, k, n_fft, hop)
                    assert p0 == edge_f and p1 >= p0 and a == edge_s and b >= a
                    if p1 > p0:                                  # the piece's frames are covered once it has landed
                        assert b == (p1 - 1) * hop + n_fft and a <= p0 * hop + (n_fft - hop if k else 0)
                    edge_f, edge_s = p1, b
                assert edge_f == f1 and edge_s == s1
    with pytest.raises(aa.AaError):
        aa.shard_piece(0, 10, 4, 4)

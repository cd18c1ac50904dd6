"""The sharded job with MORE THAN ONE RANK on the one GPU of a test box, exchanging through device memory
(aa.IpcTransport: torch IPC handles over dmabuf, interprocess events; RCCL refuses two ranks on one device and gloo would
stage every byte through the host): scatter -> compute -> gather and the overlapped schedule of aa.overlapped_e2e on
device buffers, for worlds of 2 and 3 ranks - rank 0's slab bit-equal to the unsharded launch, the schedule counted."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_samples, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    import audio_analyzer_rs_amd as aa

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = aa.IpcTransport(dist, dev)
    aa.set_transport(tr.post)
    try:
        n, hop = 1024, 256
        plan = aa.Stft(n, hop, output="mag")
        shard = aa.ShardPlan(n_samples, n, hop, world, rank)
        x = aa.signals.synth_corpus(0, n_samples)
        full = torch.from_numpy(x).to(dev) if rank == 0 else None
        whole = plan(full) if rank == 0 else None
        # serial legs
        mine = aa.scatter_samples(full, shard, device=dev, src=0)
        assert np.array_equal(mine.cpu().numpy(), x[shard.s0:shard.s1])
        rows = plan(mine)
        slab = aa.gather_rows(rows, shard, dst=0)
        tr.drain()
        if rank == 0:
            assert torch.equal(slab, whole)
        # overlapped, persistent buffers off the root, two passes (the second finds the peers' storages mapped)
        local = torch.empty(shard.local_samples, device=dev) if rank else None
        rbuf = torch.empty((shard.local_frames, plan.cols), device=dev) if rank else None
        out = torch.empty((shard.total_frames, plan.cols), device=dev) if rank == 0 else None
        for _ in range(2):
            if rank == 0:
                out.fill_(float("nan"))
            dist.barrier()
            trace = []
            aa.overlapped_e2e(full, shard, lambda loc, fb, fe, r: plan.execute_device(loc, fb, fe, out=r[fb:fe]),
                              plan.cols, n_pieces=5, src=0, device=dev, out=out, trace=trace, local=local, rows=rbuf)
            tr.drain()
            assert aa.overlap_counts(trace, 5) == {"pieces": 5, "scatter_ahead": 4, "gather_behind": 4}
            if rank == 0:
                assert torch.equal(out, whole)
        if rank == 0:
            open(os.path.join(tmpdir, "ok"), "w").write(str(tr.bytes_sent))
    finally:
        tr.close()
        aa.set_transport(None)
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_exchange_through_device_memory(tmp_path, world):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_samples = 1024 + 256 * 20000 + 77
    mp.spawn(_worker, args=(world, port, n_samples, str(tmp_path)), nprocs=world, join=True)
    sent = int(open(tmp_path / "ok").read())
    assert sent > 3 * (n_samples * 4) * (world - 1) // world          # rank 0 pushed every other rank's samples, three times

"""GPU test of the C-ABI collectives (aa_comm_*, include/aa_stft.h) on the one GPU a test box has:
a world-1 communicator on the real librccl — unique id, ncclCommInitRank, a grouped send/recv to
self (aa_comm_barrier), and the scatter / gather entry points with their degenerate world-1
meaning (a device copy into the right rows).  The multi-rank exchange pattern is the same grouped
send/recv list that the world-2/3 gloo tests (tests/test_sharding_gloo.py,
tests/test_bench_sharded_gloo.py) check through torch.distributed; RCCL refuses two ranks on one
device, so more than one rank cannot run here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aa():
    import torch  # noqa: F401  (first: the library binds to torch's HIP runtime / librccl)
    import audio_analyzer_rs_amd as aa

    return aa


def test_comm_world1_on_real_rccl(aa):
    import torch

    uid = aa.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    assert "rccl" in aa.Comm.backend()
    comm = aa.Comm(uid, 1, 0)
    try:
        comm.barrier()                                   # grouped ncclSend + ncclRecv to self
        n, hop, L = 1024, 256, 1024 + 999 * 256
        x = aa.signals.synth_corpus(0, L)
        xd = torch.from_numpy(x).cuda()
        mine = comm.scatter_samples(xd, L, n, hop, root=0)
        torch.cuda.synchronize()
        assert np.array_equal(mine.cpu().numpy(), x)
        rows = aa.Stft(n, hop, output="mag")(mine)
        slab = comm.gather_rows(rows, rows.shape[0], root=0)
        allr = comm.gather_rows(rows, rows.shape[0], root=-1)
        torch.cuda.synchronize()
        assert torch.equal(slab, rows) and torch.equal(allr, rows)
        # in place: the rank's rows already sit in its slice of the slab
        assert comm.gather_rows(slab, rows.shape[0], root=0, out=slab).data_ptr() == slab.data_ptr()
        with pytest.raises(aa.AaError):
            comm.gather_rows(rows[:-1].contiguous(), rows.shape[0], root=0)
    finally:
        comm.close()

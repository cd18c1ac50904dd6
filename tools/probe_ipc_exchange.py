"""Feasibility probe for the one-GPU rehearsal of the sharded job: two processes on ONE GPU exchange through device memory
(torch's IPC handles over dmabuf, HSA_ENABLE_IPC_MODE_LEGACY=0), ordered by an interprocess event - no host staging.
Run on a GPU box: python tools/probe_ipc_exchange.py"""
import os, socket, sys, time
import torch, torch.distributed as dist, torch.multiprocessing as mp
from torch.multiprocessing.reductions import reduce_tensor


def worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    dev = torch.device("cuda", 0)
    n = 64 << 20
    if rank == 1:                                       # the receiver owns the destination and an event
        dst = torch.zeros(n, device=dev)
        box = [reduce_tensor(dst)]
        dist.broadcast_object_list(box, src=1)
        ev_box = [None]
        dist.broadcast_object_list(ev_box, src=0)       # the sender's event
        ev = torch.cuda.Event.from_ipc_handle(dev, ev_box[0])
        flag = torch.zeros(1)
        dist.recv(flag, src=0)                          # "recorded"
        torch.cuda.current_stream().wait_event(ev)
        s = float(dst.sum().item())
        print("receiver sum", s, "expected", float(n) * 3.0, flush=True)
        dist.barrier()
    else:
        box = [None]
        dist.broadcast_object_list(box, src=1)
        fn, args = box[0]
        peer = fn(*args)                                # rank 1's memory, mapped here
        ev = torch.cuda.Event(enable_timing=False, interprocess=True)
        ev_box = [ev.ipc_handle()]
        dist.broadcast_object_list(ev_box, src=0)
        src = torch.full((n,), 3.0, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            peer.copy_(src, non_blocking=True)
            ev.record(side)
        dist.send(torch.ones(1), dst=1)
        side.synchronize()
        print(f"sender: {n * 4 / (time.perf_counter() - t0) / 1e9:.1f} GB/s incl. launch", flush=True)
        dist.barrier()
        del peer
    dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(port,), nprocs=2, join=True)

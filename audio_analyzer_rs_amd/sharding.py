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


def shard_piece(f0, f1, n_pieces, piece):
    """Frames of piece `piece` of `n_pieces` of the range [f0, f1) (aa_shard_piece: the floor rule once more)."""
    lib = _lib.load()
    p0, p1 = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.aa_shard_piece(int(f0), int(f1), int(n_pieces), int(piece), ctypes.byref(p0), ctypes.byref(p1)))
    return p0.value, p1.value


def piece_samples(f0, f1, n_pieces, piece, n_fft, hop):
    """Global sample range piece `piece` ADDS to what pieces 0 .. piece-1 of [f0, f1) delivered: consecutive pieces
    share n_fft - hop samples and every sample travels once (csrc/aa_comm.cpp: piece_samples_of)."""
    def end_of(k):
        _, q1 = shard_piece(f0, f1, n_pieces, k)
        return (q1 - 1) * hop + n_fft if q1 > f0 else f0 * hop
    return (f0 * hop if piece == 0 else end_of(piece - 1)), end_of(piece)


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
        ops = []
        for r in range(shard.world):
            f0, f1 = shard_frames(shard.total_frames, shard.world, r)
            s0, s1 = shard_samples(f0, f1, shard.n_fft, shard.hop)
            if r == src:
                local.copy_(full[s0:s1])
            elif s1 > s0:
                ops.append(("send", full[s0:s1], r))
        _p2p(ops)
    elif shard.local_samples > 0:
        _p2p([("recv", local, src)])
    return local


_TRANSPORT = None


def set_transport(post):
    """Replace torch.distributed's point-to-point calls in this module by `post(ops, group) -> requests` (ops =
    [("send" | "recv", tensor, peer)], every request has .wait()); None restores batch_isend_irecv.  Used by the
    one-GPU rehearsal of the sharded job (IpcTransport below)."""
    global _TRANSPORT
    _TRANSPORT = post


def _p2p(ops):
    """Run a list of (kind, tensor, peer) point-to-point transfers as ONE grouped batch
    (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the GPU box)."""
    import torch.distributed as dist

    if not ops:
        return
    if _TRANSPORT is not None:
        for req in _TRANSPORT(ops, None):
            req.wait()
        return
    batch = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer) for kind, t, peer in ops]
    for req in dist.batch_isend_irecv(batch):
        req.wait()


class IpcTransport:
    """Point-to-point exchanges between ranks that SHARE ONE GPU, through device memory and with device-side ordering -
    the rehearsal of the multi-GPU job on a one-GPU box (RCCL refuses two ranks on one device, and gloo would stage every
    byte through the host).  The ranks rendezvous over a host-side process group (gloo); then

      recv   the receiver publishes its destination (torch's IPC handle of the tensor: dmabuf under
             HSA_ENABLE_IPC_MODE_LEGACY=0) to the sender;
      send   the sender maps it, copies device-to-device on a side stream behind its own compute stream, records an
             INTERPROCESS event behind the copy and tells the receiver which one;
      wait   the receiver lets its compute stream wait for that event - no host synchronisation, no host staging.

    Sends and receives of a pair match in posting order.  Asynchronous like batch_isend_irecv: a posted send returns as
    soon as its copy is queued.  `drain()` - after the device is idle - releases the mapped peer buffers and the host
    messages; call it between passes.  (The bytes never leave the one GPU: what this rehearses is the SCHEDULE of
    aa.overlapped_e2e on device buffers, event for event, not a link.)"""

    TAG_DESC, TAG_HANDLE, TAG_DONE = 11, 12, 13

    def __init__(self, dist, device, n_events=256):
        import torch

        self.dist, self.dev, self.torch = dist, device, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.side = torch.cuda.Stream(device)
        self.events = [torch.cuda.Event(enable_timing=False, interprocess=True) for _ in range(n_events)]
        for e in self.events:
            e.record(self.side)                      # (an event must have been recorded before its handle is taken)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, [e.ipc_handle() for e in self.events])
        self.peer_events = {r: [torch.cuda.Event.from_ipc_handle(device, h) for h in hs]
                            for r, hs in enumerate(gathered) if r != self.rank}
        self.seq, self.works, self.keep = 0, [], []
        self.announced = {}                          # (peer, storage address, bytes) -> id: handles this rank has sent
        self.opened = {}                             # (peer, id) -> the peer's storage as a flat float32 tensor
        self.bytes_sent = 0

    def post(self, ops, group=None):
        import pickle
        from torch.multiprocessing.reductions import reduce_tensor

        torch, dist = self.torch, self.dist
        reqs = []
        for kind, t, peer in ops:                    # receives first: the senders wait for these descriptors
            if kind != "recv":
                continue
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise ValueError("IpcTransport moves contiguous CUDA float32 tensors")
            st = t.untyped_storage()
            key = (peer, st.data_ptr(), st.nbytes())
            first = key not in self.announced
            if first:
                self.announced[key] = len(self.announced)
            handle = b""
            if first:                                # the whole storage once; later receives into it name it by id
                flat = torch.empty(0, dtype=torch.float32, device=t.device).set_(st, 0, (st.nbytes() // 4,))
                handle = pickle.dumps(reduce_tensor(flat))
            desc = torch.tensor([self.announced[key], t.storage_offset(), t.numel(), len(handle)], dtype=torch.int64)
            self.works.append(dist.isend(desc, peer, tag=self.TAG_DESC))
            self.keep += [desc, t]
            if first:
                buf = torch.frombuffer(bytearray(handle), dtype=torch.uint8)
                self.works.append(dist.isend(buf, peer, tag=self.TAG_HANDLE))
                self.keep.append(buf)
            reqs.append(_IpcRecv(self, peer))
        for kind, t, peer in ops:
            if kind != "send":
                continue
            desc = torch.zeros(4, dtype=torch.int64)
            dist.recv(desc, peer, tag=self.TAG_DESC)
            sid, off, numel, hlen = (int(v) for v in desc.tolist())
            if hlen:
                buf = torch.empty(hlen, dtype=torch.uint8)
                dist.recv(buf, peer, tag=self.TAG_HANDLE)
                fn, args = pickle.loads(buf.numpy().tobytes())
                self.opened[(peer, sid)] = fn(*args)                 # the receiver's storage, mapped into this process
            dst = self.opened[(peer, sid)][off:off + numel]
            if numel != t.numel():
                raise ValueError(f"send of {t.numel()} elements meets a receive of {numel}")
            self.side.wait_stream(torch.cuda.current_stream(self.dev))      # behind whatever produced t
            slot = self.seq % len(self.events)
            self.seq += 1
            with torch.cuda.stream(self.side):
                dst.copy_(t.reshape(-1), non_blocking=True)
                self.events[slot].record(self.side)
            note = torch.tensor([slot], dtype=torch.int64)
            self.works.append(dist.isend(note, peer, tag=self.TAG_DONE))
            self.keep += [note, t]
            self.bytes_sent += t.numel() * t.element_size()
            reqs.append(_IpcSent())
        return reqs

    def drain(self):
        """With the device idle: complete the host messages and let go of what was held for them (the peers' storages
        stay mapped: the next pass finds them by id - which is why destinations should be buffers that live across
        passes: a shared block goes back to its owner's allocator only when every peer has let go of it)."""
        self.torch.cuda.synchronize(self.dev)
        for w in self.works:
            w.wait()
        self.works, self.keep = [], []

    def close(self):
        """Unmap the peers' storages (before the owners exit) and leave together."""
        self.drain()
        self.opened.clear()
        self.dist.barrier()


class _IpcRecv:
    def __init__(self, tr, peer):
        self.tr, self.peer = tr, peer

    def wait(self):
        tr = self.tr
        note = tr.torch.zeros(1, dtype=tr.torch.int64)
        tr.dist.recv(note, self.peer, tag=tr.TAG_DONE)           # "copy and event are queued"
        tr.torch.cuda.current_stream(tr.dev).wait_event(tr.peer_events[self.peer][int(note.item())])


class _IpcSent:
    def wait(self):
        pass


def gather_rows(rows, shard, dst=None, out=None):
    """rows: this rank's [local_frames, cols] tensor.
    dst=r    -> rank r returns [F, cols] (written into `out` if given), the others None.  Every
                rank sends its own UNPADDED block straight into its slice of the destination
                (grouped point-to-point: what lands on the root is exactly F*cols*4 bytes).
    dst=None -> all ranks get [F, cols] (all_gather on blocks padded to the largest shard)."""
    import torch
    import torch.distributed as dist

    if shard.world == 1:
        if out is not None:
            out.copy_(rows)
            return out
        return rows
    cols = rows.shape[1]
    ranges = shard.all_frame_ranges()
    if dst is not None:
        if shard.rank == dst:
            if out is None:
                out = torch.empty((shard.total_frames, cols), dtype=rows.dtype, device=rows.device)
            f0, f1 = ranges[dst]
            out[f0:f1].copy_(rows)
            _p2p([("recv", out[g0:g1], r) for r, (g0, g1) in enumerate(ranges) if r != dst and g1 > g0])
            return out
        if rows.shape[0] > 0:
            _p2p([("send", rows.contiguous(), dst)])
        return None
    cap = max(f1 - f0 for f0, f1 in ranges)
    pad = torch.zeros((cap, cols), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    blocks = [torch.empty_like(pad) for _ in range(shard.world)]
    dist.all_gather(blocks, pad)
    return torch.cat([b[:f1 - f0] for b, (f0, f1) in zip(blocks, ranges)], dim=0)


def overlapped_e2e(full, shard, execute, cols, n_pieces=8, src=0, device=None, out=None, group_in=None, group_out=None,
                   trace=None, post=None, local=None, rows=None):
    """scatter -> compute -> gather of a whole job with the three legs overlapped, over torch.distributed: every
    rank's frame range is cut into `n_pieces` pieces (shard_piece) and piece k + 1's samples travel while piece k
    is transformed and piece k - 1's rows go back to `src`.

      full     rank `src`: the whole sample buffer (1-D float32); ignored elsewhere
      execute  execute(local, frame_begin, frame_end, rows): queue the transform of local frames [frame_begin,
               frame_end) of `local` (the rank's sample range) into rows[frame_begin:frame_end]
      out      rank `src`: optional [total_frames, cols] destination
    Returns (local, rows, out): out is the dense slab on `src`, None elsewhere; on `src` local / rows are views
    of full / out (nothing is copied).

    The exchanges are posted asynchronously (batch_isend_irecv) and waited for only where their data is needed; on
    the "nccl" backend a wait() orders the current stream behind the exchange, and an exchange posted after a
    launch is ordered behind it, so compute k overlaps scatter k + 1 and gather k - 1 on the device.  Every rank
    posts S0, S1, G0, S2, G1, .. in this order (one communicator serialises them in issue order; pass two groups
    to let scatters and gathers run side by side).

      trace    optional list: the schedule as it was issued - ("S", k) scatter k posted, ("W", k) piece k's samples
               waited for, ("C", k) compute k queued (empty pieces included), ("G", k) gather k posted, ("J",) the
               gathers joined; `overlap_counts(trace)` says how many computes were queued with an exchange in flight
      post     optional transport: post(ops, group) -> requests with .wait(), ops = [("send" | "recv", tensor, peer)];
               default: torch.distributed.batch_isend_irecv
      local, rows  off `src`: optional buffers of the rank's sample range / rows to use instead of fresh ones (a
               transport that maps the destinations into the sender, like IpcTransport, then maps them once)"""
    import torch
    import torch.distributed as dist

    P = int(n_pieces)
    dev = device if device is not None else (full.device if full is not None else "cpu")
    root = shard.rank == src
    ranges = shard.all_frame_ranges()
    if root:
        if out is None:
            out = torch.empty((shard.total_frames, cols), dtype=torch.float32, device=dev)
        local = full[shard.s0:shard.s1]
        rows = out[shard.f0:shard.f1]
    else:
        if local is None:
            local = torch.empty(shard.local_samples, dtype=torch.float32, device=dev)
        if rows is None:
            rows = torch.empty((shard.local_frames, cols), dtype=torch.float32, device=dev)
        if local.numel() != shard.local_samples or tuple(rows.shape) != (shard.local_frames, cols):
            raise ValueError("local / rows do not have the rank's shapes")

    def post_default(ops, group):
        if not ops or shard.world == 1:
            return []
        batch = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer, group) for kind, t, peer in ops]
        return dist.batch_isend_irecv(batch)

    if post is None:
        post = _TRANSPORT if _TRANSPORT is not None else post_default
    note = trace.append if trace is not None else (lambda entry: None)

    def scatter(k):
        ops = []
        if root:
            for r, (g0, g1) in enumerate(ranges):
                a, b = piece_samples(g0, g1, P, k, shard.n_fft, shard.hop)
                if r != src and b > a:
                    ops.append(("send", full[a:b], r))
        else:
            a, b = piece_samples(shard.f0, shard.f1, P, k, shard.n_fft, shard.hop)
            if b > a:
                ops.append(("recv", local[a - shard.s0:b - shard.s0], src))
        return post(ops, group_in)

    def gather(k):
        ops = []
        if root:
            for r, (g0, g1) in enumerate(ranges):
                q0, q1 = shard_piece(g0, g1, P, k)
                if r != src and q1 > q0:
                    ops.append(("recv", out[q0:q1], r))
        else:
            q0, q1 = shard_piece(shard.f0, shard.f1, P, k)
            if q1 > q0:
                ops.append(("send", rows[q0 - shard.f0:q1 - shard.f0], src))
        return post(ops, group_out)

    pending = {0: scatter(0)}
    note(("S", 0))
    tail = []
    for k in range(P):
        if k + 1 < P:
            pending[k + 1] = scatter(k + 1)
            note(("S", k + 1))
        for req in pending.pop(k):
            req.wait()
        note(("W", k))
        q0, q1 = shard_piece(shard.f0, shard.f1, P, k)
        if q1 > q0:
            execute(local, q0 - shard.f0, q1 - shard.f0, rows)
        note(("C", k))
        tail += gather(k)
        note(("G", k))
    for req in tail:
        req.wait()
    note(("J",))
    return local, rows, (out if root else None)


def overlap_counts(trace, n_pieces):
    """What a trace of overlapped_e2e says about the schedule, by counting (not timing): of the n_pieces computes, how
    many were queued while the NEXT piece's scatter was already posted and not yet waited for (`scatter_ahead`), and how
    many while an earlier piece's gather was posted and not yet joined (`gather_behind`) - n_pieces - 1 each when the
    three legs overlap as designed, 0 each in the serial leg.  Also checks the issue order S0, S1, [W0 C0] G0, S2,
    [W1 C1] G1, .. and raises ValueError if the trace departs from it."""
    P = int(n_pieces)
    want = [("S", 0)]
    for k in range(P):
        if k + 1 < P:
            want.append(("S", k + 1))
        want += [("W", k), ("C", k), ("G", k)]
    want.append(("J",))
    if list(trace) != want:
        raise ValueError(f"schedule departs from S0 S1 W0 C0 G0 S2 ..: {list(trace)[:12]} ..")
    posted_s, waited_s, posted_g, joined = set(), set(), set(), False
    scatter_ahead = gather_behind = 0
    for entry in trace:
        kind = entry[0]
        if kind == "S":
            posted_s.add(entry[1])
        elif kind == "W":
            waited_s.add(entry[1])
        elif kind == "G":
            posted_g.add(entry[1])
        elif kind == "J":
            joined = True
        elif kind == "C":
            k = entry[1]
            scatter_ahead += 1 if (k + 1) in posted_s and (k + 1) not in waited_s else 0
            gather_behind += 1 if any(g < k for g in posted_g) and not joined else 0
    return {"pieces": P, "scatter_ahead": scatter_ahead, "gather_behind": gather_behind}


def gather_samples(local, shard, dst=0):
    """Inverse of scatter_samples: rank `dst` assembles the whole buffer from every rank's
    [s0, s1) range (the n_fft-hop halos overlap; each rank contributes only its new samples)."""
    import torch

    ranges = []
    for r in range(shard.world):
        f0, f1 = shard_frames(shard.total_frames, shard.world, r)
        ranges.append(shard_samples(f0, f1, shard.n_fft, shard.hop))
    if shard.world == 1:
        return local
    if shard.rank == dst:
        full = torch.zeros(shard.n_samples, dtype=local.dtype, device=local.device)
        s0, s1 = ranges[dst]
        full[s0:s1].copy_(local)
        _p2p([("recv", full[a:b], r) for r, (a, b) in enumerate(ranges) if r != dst and b > a])
        return full
    if local.numel() > 0:
        _p2p([("send", local.contiguous(), dst)])
    return None


class Comm:
    """The C-ABI collectives (`aa_comm_*`, include/aa_stft.h) — what a Rust / C++ host calls; RCCL
    over xGMI, bound by libaa_stft.so itself (no torch.distributed involved).

        uid = Comm.unique_id() on rank 0, handed to the other ranks (128 bytes, any channel)
        comm = Comm(uid, world, rank)             # after torch.cuda.set_device(local_rank)
        mine = comm.scatter_samples(full, n_samples, n_fft, hop, root=0)
        rows = comm.gather_rows(my_rows, total_frames, root=0)
    """

    def __init__(self, uid, world, rank):
        self._lib = _lib.load()
        if len(uid) != 128:
            raise _lib.AaError(_lib.AA_ERR_ARG, "a communicator id is 128 bytes")
        self._uid = ctypes.create_string_buffer(bytes(uid), 128)
        self._comm = ctypes.c_void_p()
        _lib.check(self._lib.aa_comm_create(self._uid, int(world), int(rank), ctypes.byref(self._comm)))
        self.world, self.rank = int(world), int(rank)

    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.aa_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def backend():
        return _lib.load().aa_comm_backend().decode()

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.aa_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream(t, stream):
        import torch

        return stream if stream is not None else torch.cuda.current_stream(t.device).cuda_stream

    def barrier(self, stream=None):
        import torch

        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(self._lib.aa_comm_barrier(self._comm, st))

    def scatter_samples(self, full, n_samples, n_fft, hop, root=0, stream=None):
        """full: CUDA float32 [n_samples] on rank `root` (None elsewhere).  Returns this rank's range."""
        import torch

        shard = ShardPlan(n_samples, n_fft, hop, self.world, self.rank)
        local = torch.empty(shard.local_samples, dtype=torch.float32, device="cuda")
        if self.rank == root and not (full.is_cuda and full.dtype == torch.float32 and full.is_contiguous()
                                      and full.numel() == n_samples):
            raise _lib.AaError(_lib.AA_ERR_ARG, "full must be a contiguous CUDA float32 tensor of n_samples")
        _lib.check(self._lib.aa_scatter_samples(
            self._comm, full.data_ptr() if self.rank == root else None, int(n_samples), int(n_fft), int(hop),
            int(root), local.data_ptr() if local.numel() else None, local.numel(), self._stream(local, stream)))
        return local

    def execute_sharded(self, plan, full, n_samples, root=0, n_pieces=8, comm_out=None, out=None, stream=None,
                        alias=True):
        """aa_stft_execute_sharded: rank `root`'s samples in, the dense [total_frames, cols] slab on `root` out, with
        scatter k + 1 || compute k || gather k - 1 (pieces of every rank's range).  `comm_out`: a second communicator
        for the gathers (they then run next to the scatters instead of taking turns with them).  Stream-ordered on
        `stream`; returns (local, rows, out) - out None off the root; on the root local / rows are views of full /
        out (alias=False: buffers of their own, filled by device copies on the exchange streams)."""
        import torch

        shard = ShardPlan(n_samples, plan.n_fft, plan.hop, self.world, self.rank)
        cols = plan.cols
        if self.rank == root:
            if not (full.is_cuda and full.dtype == torch.float32 and full.is_contiguous() and full.numel() == n_samples):
                raise _lib.AaError(_lib.AA_ERR_ARG, "full must be a contiguous CUDA float32 tensor of n_samples")
            if out is None:
                out = torch.empty((shard.total_frames, cols), dtype=torch.float32, device=full.device)
            local, rows = full[shard.s0:shard.s1], out[shard.f0:shard.f1]
            if not alias:
                local, rows = torch.empty_like(local), torch.empty_like(rows)
        else:
            local = torch.empty(shard.local_samples, dtype=torch.float32, device="cuda")
            rows = torch.empty((shard.local_frames, cols), dtype=torch.float32, device="cuda")
            out = None
        ptr = lambda t: t.data_ptr() if t is not None and t.numel() else None
        _lib.check(self._lib.aa_stft_execute_sharded(
            plan._plan, self._comm, (comm_out or self)._comm, ptr(full) if self.rank == root else None, int(n_samples),
            int(root), ptr(local), local.numel(), ptr(rows), ptr(out), int(n_pieces), self._stream(rows, stream)))
        return local, rows, out

    def gather_rows(self, rows, total_frames, root=0, out=None, stream=None):
        """rows: this rank's CUDA float32 [local_frames, cols].  Rank `root` (every rank for
        root=-1) returns the dense [total_frames, cols] slab, the others None."""
        import torch

        if not (rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 2 and rows.is_contiguous()):
            raise _lib.AaError(_lib.AA_ERR_ARG, "rows must be a contiguous 2-D CUDA float32 tensor")
        f0, f1 = shard_frames(total_frames, self.world, self.rank)
        if rows.shape[0] != f1 - f0:
            raise _lib.AaError(_lib.AA_ERR_LEN, f"rank {self.rank} owns {f1 - f0} frames, rows has {rows.shape[0]}")
        cols = rows.shape[1]
        receiver = root < 0 or root == self.rank
        if receiver:
            if out is None:
                out = torch.empty((total_frames, cols), dtype=torch.float32, device=rows.device)
            elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
                      and out.numel() >= total_frames * cols):
                raise _lib.AaError(_lib.AA_ERR_ARG, "out must be a contiguous CUDA float32 [total_frames, cols]")
        _lib.check(self._lib.aa_gather_rows(self._comm, rows.data_ptr() if rows.numel() else None,
                                            int(total_frames), int(cols), int(root),
                                            out.data_ptr() if receiver else None, self._stream(rows, stream)))
        return out if receiver else None

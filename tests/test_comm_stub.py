"""Multi-rank call lists of the C-ABI collectives (csrc/aa_comm.cpp: aa_scatter_samples, aa_gather_rows,
aa_comm_barrier) for worlds of 2, 3 and 8 ranks, without a second GPU: a loopback stand-in for librccl
(tests/stubs/rccl_loopback.c, bound through AA_RCCL_PATH) keeps every rank of a world in ONE process, logs
each ncclSend / ncclRecv (rank, peer, pointer, count, group, current device) and carries matched pairs out.
Checked from every rank, for roots at both ends and the all-gather form, for frame totals below the world
size and for an empty job: sends and receives pair up, their offsets and counts are those of
aa_shard_frames / aa_shard_samples, the bytes arrive, no group is left open on an error path and the
communicator's device is selected around every call.

CPU leg (`-m "not gpu"`): the few HIP calls of aa_comm.cpp are answered from host memory by
tests/stubs/hip_host.c (LD_PRELOAD in a child process that does not import torch).  GPU leg (`-m gpu`): the
same child on the real HIP runtime with device buffers.  What this cannot show is RCCL itself moving the
bytes between two GPUs: that needs the driver's multi-GPU node."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "stubs")

CHILD = r'''
import ctypes, os, sys
import numpy as np

ROOT, RCCL_STUB, HOST_HIP = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
assert "torch" not in sys.modules
if not HOST_HIP:
    ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
hip = ctypes.CDLL(None)                     # global scope: the preloaded stand-in first, else the real runtime
lib = ctypes.CDLL(os.path.join(ROOT, "audio_analyzer_rs_amd", "libaa_stft.so"))
stub = ctypes.CDLL(RCCL_STUB)               # same handle the library dlopens (one copy per path)
vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t

hip.hipMalloc.argtypes = [ctypes.POINTER(vp), sz]
hip.hipMemcpy.argtypes = [vp, vp, sz, ctypes.c_int]
hip.hipFree.argtypes = [vp]
hip.hipSetDevice.argtypes = [ctypes.c_int]
lib.aa_last_error.restype = ctypes.c_char_p
lib.aa_comm_backend.restype = ctypes.c_char_p
lib.aa_comm_create.argtypes = [vp, i32, i32, ctypes.POINTER(vp)]
lib.aa_comm_destroy.argtypes = [vp]
lib.aa_comm_barrier.argtypes = [vp, vp]
lib.aa_scatter_samples.argtypes = [vp, vp, i64, i32, i32, i32, vp, i64, vp]
lib.aa_gather_rows.argtypes = [vp, vp, i64, i32, i32, vp, vp]
lib.aa_shard_frames.argtypes = [i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
lib.aa_shard_samples.argtypes = [i64, i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
stub.aa_stub_log_get.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 5 + [ctypes.POINTER(ctypes.c_size_t)] * 2
stub.aa_stub_log_stream.argtypes = [ctypes.c_int]
stub.aa_stub_log_stream.restype = ctypes.c_size_t
PIECE_FN = ctypes.CFUNCTYPE(ctypes.c_int, vp, i64, i64, vp)
lib.aa_shard_piece.argtypes = [i64, i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
lib.aa_scatter_samples_piece.argtypes = [vp, vp, i64, i32, i32, i32, vp, i64, i32, i32, vp]
lib.aa_gather_rows_piece.argtypes = [vp, vp, i64, i32, i32, vp, i32, i32, vp]
lib.aa_sharded_pipeline.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp, vp, i32, PIECE_FN, vp, vp]
hip.hipDeviceSynchronize.argtypes = []
if HOST_HIP:
    hip.aa_hip_host_sync_get.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t),
                                         ctypes.POINTER(ctypes.c_size_t)]


def ok(rc):
    assert rc == 0, (rc, lib.aa_last_error())


class Dev:
    """n floats of `device` memory (host memory under the stand-in)."""
    def __init__(self, n, fill=None):
        self.n = n
        p = vp()
        assert hip.hipMalloc(ctypes.byref(p), max(4 * n, 4)) == 0
        self.ptr = p.value
        if fill is not None:
            self.put(np.ascontiguousarray(fill, dtype=np.float32))
        else:
            self.put(np.full(n, np.nan, dtype=np.float32))
    def put(self, a):
        assert a.size == self.n
        if self.n:
            assert hip.hipMemcpy(self.ptr, a.ctypes.data, 4 * self.n, 1) == 0
    def get(self):
        a = np.empty(self.n, dtype=np.float32)
        if self.n:
            assert hip.hipMemcpy(a.ctypes.data, self.ptr, 4 * self.n, 2) == 0
        return a
    def free(self):
        hip.hipFree(self.ptr)


def shard_frames(total, world, r):
    f0, f1 = i64(), i64()
    ok(lib.aa_shard_frames(total, world, r, ctypes.byref(f0), ctypes.byref(f1)))
    return f0.value, f1.value


def shard_samples(f0, f1, n, hop):
    s0, s1 = i64(), i64()
    ok(lib.aa_shard_samples(f0, f1, n, hop, ctypes.byref(s0), ctypes.byref(s1)))
    return s0.value, s1.value


def log():
    out = []
    for i in range(stub.aa_stub_log_count()):
        v = [ctypes.c_int() for _ in range(5)]
        p, c = ctypes.c_size_t(), ctypes.c_size_t()
        assert stub.aa_stub_log_get(i, *[ctypes.byref(x) for x in v], ctypes.byref(p), ctypes.byref(c)) == 0
        out.append(dict(op="send" if v[0].value == 0 else "recv", rank=v[1].value, peer=v[2].value,
                        in_group=v[3].value, device=v[4].value, ptr=p.value, count=c.value,
                        stream=stub.aa_stub_log_stream(i)))
    return out


def sync_log():
    """(kind, stream, event) of every hipEventRecord ("rec") / hipStreamWaitEvent ("wait") so far (host stand-in only)"""
    out = []
    for i in range(hip.aa_hip_host_sync_count()):
        k, st, ev = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.aa_hip_host_sync_get(i, ctypes.byref(k), ctypes.byref(st), ctypes.byref(ev)) == 0
        out.append(("rec" if k.value == 0 else "wait", st.value, ev.value))
    return out


def shard_piece(f0, f1, P, k):
    p0, p1 = i64(), i64()
    ok(lib.aa_shard_piece(f0, f1, P, k, ctypes.byref(p0), ctypes.byref(p1)))
    return p0.value, p1.value


def piece_samples(f0, f1, P, k, n, hop):
    """the samples piece k adds to pieces 0 .. k-1 (every sample travels once)"""
    def end_of(kk):
        q1 = shard_piece(f0, f1, P, kk)[1]
        return (q1 - 1) * hop + n if q1 > f0 else f0 * hop
    return (f0 * hop if k == 0 else end_of(k - 1)), end_of(k)


def quiet():
    assert stub.aa_stub_errors() == 0 and stub.aa_stub_pending() == 0 and stub.aa_stub_group_depth() == 0
    a, b = ctypes.c_int(), ctypes.c_int()
    stub.aa_stub_group_calls(ctypes.byref(a), ctypes.byref(b))
    assert a.value == b.value, "ncclGroupStart / ncclGroupEnd calls differ"


def pairs_up(entries):
    """every send (r -> p, count) has its receive (p <- r, count), in order per pair of ranks"""
    sends, recvs = {}, {}
    for e in entries:
        key = (e["rank"], e["peer"]) if e["op"] == "send" else (e["peer"], e["rank"])
        (sends if e["op"] == "send" else recvs).setdefault(key, []).append(e["count"])
    assert sends == recvs, (sends, recvs)


assert b"AA_RCCL_PATH" in lib.aa_comm_backend(), lib.aa_comm_backend()
uid = (ctypes.c_char * 128)()
ok(lib.aa_comm_unique_id(uid))
N, HOP = 1024, 256
checked = 0
for world in (2, 3, 8):
    comms = []
    for r in range(world):
        if HOST_HIP:
            assert hip.hipSetDevice(r) == 0          # rank r's GPU is current when its communicator is made
        c = vp()
        ok(lib.aa_comm_create(uid, world, r, ctypes.byref(c)))
        comms.append(c)
    if HOST_HIP:
        assert hip.hipSetDevice(0) == 0              # ... and is NOT current any more when it is used
    dev_of = (lambda r: r) if HOST_HIP else (lambda r: 0)

    # ---- barrier: world sends + world receives per rank, one group each ----
    stub.aa_stub_reset()
    for r in range(world):
        ok(lib.aa_comm_barrier(comms[r], None))
    L = log()
    quiet(); pairs_up(L)
    assert len(L) == 2 * world * world and all(e["in_group"] and e["count"] == 1 for e in L)
    assert all(e["device"] == dev_of(e["rank"]) for e in L)

    for total_frames in (1000 + 37 * world + 5, world - 1, 1, 0):
        n_samples = 0 if total_frames == 0 else N + (total_frames - 1) * HOP + 17     # a ragged tail too
        x = np.random.default_rng(world * 1000 + total_frames).standard_normal(n_samples).astype(np.float32)
        ranges = [shard_samples(*shard_frames(total_frames, world, r), N, HOP) for r in range(world)]
        assert sum(f1 - f0 for f0, f1 in (shard_frames(total_frames, world, r) for r in range(world))) == total_frames
        for root in (0, world - 1):
            # ---- scatter ----
            stub.aa_stub_reset()
            full = Dev(n_samples, x)
            local = [Dev(s1 - s0 + 3) for s0, s1 in ranges]
            for r in range(world):                                   # root first or root last: both orders
                ok(lib.aa_scatter_samples(comms[r], full.ptr if r == root else None, n_samples, N, HOP, root,
                                          local[r].ptr, local[r].n, None))
            L = log()
            quiet(); pairs_up(L)
            for r, (s0, s1) in enumerate(ranges):
                got = local[r].get()
                assert np.array_equal(got[:s1 - s0], x[s0:s1]) and np.isnan(got[s1 - s0:]).all(), (world, root, r)
            sends = [e for e in L if e["op"] == "send"]
            want = {r: ranges[r] for r in range(world) if r != root and ranges[r][1] > ranges[r][0]}
            assert {e["peer"]: ((e["ptr"] - full.ptr) // 4, (e["ptr"] - full.ptr) // 4 + e["count"]) for e in sends} == want
            assert all(e["rank"] == root and e["in_group"] and e["device"] == dev_of(root) for e in sends)
            recvs = [e for e in L if e["op"] == "recv"]
            assert {e["rank"]: (e["ptr"], e["count"]) for e in recvs} == \
                {r: (local[r].ptr, s1 - s0) for r, (s0, s1) in want.items()}
            assert all(e["peer"] == root and e["device"] == dev_of(e["rank"]) for e in recvs)
            # a local buffer one sample short is refused before anything is posted
            if ranges[root ^ 1 if world == 2 else (root + 1) % world][1] > 0:
                r = (root + 1) % world
                stub.aa_stub_reset()
                rc = lib.aa_scatter_samples(comms[r], None, n_samples, N, HOP, root, local[r].ptr,
                                            ranges[r][1] - ranges[r][0] - 1, None)
                assert rc == -2 and stub.aa_stub_log_count() == 0, rc        # AA_ERR_LEN
            for b in local + [full]:
                b.free()
            checked += 1
        for cols in (5, 513):
            rows_np = [np.arange(f0 * cols, f1 * cols, dtype=np.float32) for f0, f1 in
                       (shard_frames(total_frames, world, r) for r in range(world))]
            for root in (0, world - 1, -1):
                # ---- gather (root = -1: every rank receives every block) ----
                stub.aa_stub_reset()
                rows = [Dev(a.size, a) for a in rows_np]
                outs = [Dev(total_frames * cols) if root in (-1, r) else None for r in range(world)]
                for r in range(world):
                    ok(lib.aa_gather_rows(comms[r], rows[r].ptr if rows[r].n else None, total_frames, cols, root,
                                          outs[r].ptr if outs[r] and outs[r].n else None, None))
                L = log()
                quiet(); pairs_up(L)
                expect = np.arange(total_frames * cols, dtype=np.float32)
                for r in range(world):
                    if outs[r] is not None:
                        assert np.array_equal(outs[r].get(), expect), (world, root, r, cols)
                for e in L:
                    assert e["in_group"] and e["device"] == dev_of(e["rank"])
                    f0, f1 = shard_frames(total_frames, world, e["rank"] if e["op"] == "send" else e["peer"])
                    assert e["count"] == (f1 - f0) * cols and e["count"] > 0
                    if e["op"] == "send":
                        assert e["ptr"] == rows[e["rank"]].ptr and (root < 0 or e["peer"] == root)
                    else:
                        assert e["ptr"] == outs[e["rank"]].ptr + 4 * f0 * cols
                nonempty = [r for r in range(world) if rows_np[r].size]
                receivers = list(range(world)) if root < 0 else [root]
                assert len([e for e in L if e["op"] == "send"]) == sum(1 for s in nonempty for d in receivers if s != d)
                # error path: rows missing on a rank that owns frames -> AA_ERR_ARG, no group left open
                if nonempty:
                    stub.aa_stub_reset()
                    assert lib.aa_gather_rows(comms[nonempty[0]], None, total_frames, cols, root, None, None) == -1
                    assert stub.aa_stub_group_depth() == 0 and stub.aa_stub_log_count() == 0
                for b in rows + [o for o in outs if o is not None]:
                    b.free()
                checked += 1
    # ---- the exchanges in pieces, and the overlapped pipeline (scatter k + 1 || compute k || gather k - 1) ----
    comms2 = []
    for r in range(world):
        if HOST_HIP:
            assert hip.hipSetDevice(r) == 0
        c = vp()
        ok(lib.aa_comm_create(uid, world, r, ctypes.byref(c)))
        comms2.append(c)
    if HOST_HIP:
        assert hip.hipSetDevice(0) == 0
    cols = 7
    for total_frames, P, root in ((1000 + 37 * world + 5, 8, 0), (1000 + 37 * world + 5, 5, world - 1), (world + 1, 8, 0),
                                  (world - 1, 3, world - 1), (0, 4, 0), (997, 1, 0)):
        n_samples = 0 if total_frames == 0 else N + (total_frames - 1) * HOP + 17
        x = np.random.default_rng(world * 77 + total_frames + P).standard_normal(n_samples).astype(np.float32)
        frames = [shard_frames(total_frames, world, r) for r in range(world)]
        ranges = [shard_samples(f0, f1, N, HOP) for f0, f1 in frames]
        order = [root] + [r for r in range(world) if r != root]      # the root's sends first: a matched pair is carried
                                                                     # out when its second half is posted
        # (A) scatter in pieces: every sample once, at its place in the rank's buffer
        stub.aa_stub_reset()
        full = Dev(n_samples, x)
        local = [Dev(s1 - s0 + 3) for s0, s1 in ranges]
        for k in range(P):
            for r in order:
                ok(lib.aa_scatter_samples_piece(comms[r], full.ptr if r == root else None, n_samples, N, HOP, root,
                                                local[r].ptr, local[r].n, P, k, None))
        L = log()
        quiet(); pairs_up(L)
        for r, (s0, s1) in enumerate(ranges):
            got = local[r].get()
            assert np.array_equal(got[:s1 - s0], x[s0:s1]) and np.isnan(got[s1 - s0:]).all(), (world, root, r, P)
        want_sends = [(r, *piece_samples(*frames[r], P, k, N, HOP)) for k in range(P) for r in range(world) if r != root]
        want_sends = [(r, a, b) for r, a, b in want_sends if b > a]
        assert [(e["peer"], (e["ptr"] - full.ptr) // 4, (e["ptr"] - full.ptr) // 4 + e["count"]) for e in L if e["op"] == "send"] == want_sends
        assert [(e["rank"], (e["ptr"] - local[e["rank"]].ptr) // 4 + ranges[e["rank"]][0],
                 (e["ptr"] - local[e["rank"]].ptr) // 4 + ranges[e["rank"]][0] + e["count"]) for e in L if e["op"] == "recv"] == want_sends
        for r in range(world):                                       # per rank the pieces add up to its range, nothing twice
            assert sum(b - a for rr, a, b in want_sends if rr == r) == (ranges[r][1] - ranges[r][0] if r != root else 0)
        # (B) gather in pieces
        stub.aa_stub_reset()
        rows_np = [np.arange(f0 * cols, f1 * cols, dtype=np.float32) for f0, f1 in frames]
        rows = [Dev(a.size, a) for a in rows_np]
        out = Dev(total_frames * cols)
        for k in range(P):
            for r in order[::-1]:                                    # receives first this time
                ok(lib.aa_gather_rows_piece(comms[r], rows[r].ptr if rows[r].n else None, total_frames, cols, root,
                                            out.ptr if r == root and out.n else None, P, k, None))
        L = log()
        quiet(); pairs_up(L)
        assert np.array_equal(out.get(), np.arange(total_frames * cols, dtype=np.float32))
        want = [(r, *shard_piece(*frames[r], P, k)) for k in range(P) for r in range(world) if r != root]
        want = [(r, q0, q1) for r, q0, q1 in want if q1 > q0]
        assert sorted((e["rank"], (e["ptr"] - rows[e["rank"]].ptr) // (4 * cols) + frames[e["rank"]][0], e["count"] // cols)
                      for e in L if e["op"] == "send") == sorted((r, q0, q1 - q0) for r, q0, q1 in want)
        assert [(e["peer"], (e["ptr"] - out.ptr) // (4 * cols), (e["ptr"] - out.ptr) // (4 * cols) + e["count"] // cols)
                for e in L if e["op"] == "recv"] == want
        for b in local + rows + [out]:
            b.free()
        # (C) the pipeline, on one communicator and on two
        for two in (False, True):
            stub.aa_stub_reset()
            if HOST_HIP:
                hip.aa_hip_host_sync_reset()
            USER = 7 if HOST_HIP else None                           # the caller's stream (the host stand-in takes any value)
            local = [Dev(s1 - s0 + 3) for s0, s1 in ranges]
            rows = [Dev((f1 - f0) * cols) for f0, f1 in frames]
            out = Dev(total_frames * cols)
            calls = {r: [] for r in range(world)}
            spans = {}
            for r in order:
                def compute(user, fb, fe, stream, r=r):
                    # rows[i][c] = 2 local[i hop + 3 c] + 1 for the piece's frames: right only if the piece's samples have landed
                    hip.hipDeviceSynchronize()
                    calls[r].append((fb, fe, stream, stub.aa_stub_log_count()))
                    loc = local[r].get()
                    part = np.stack([2.0 * loc[i * HOP + 3 * np.arange(cols)] + 1.0 for i in range(fb, fe)]).astype(np.float32)
                    assert hip.hipMemcpy(rows[r].ptr + 4 * fb * cols, part.ctypes.data, 4 * part.size, 1) == 0
                    return 0
                cb = PIECE_FN(compute)
                n0 = stub.aa_stub_log_count()
                m0 = hip.aa_hip_host_sync_count() if HOST_HIP else 0
                ok(lib.aa_sharded_pipeline(comms[r], (comms2 if two else comms)[r], full.ptr if r == root else None, n_samples,
                                           N, HOP, cols, root, local[r].ptr, local[r].n, rows[r].ptr if rows[r].n else None,
                                           out.ptr if r == root and out.n else None, P, cb, None, USER))
                spans[r] = (n0, stub.aa_stub_log_count(), m0, hip.aa_hip_host_sync_count() if HOST_HIP else 0)
            hip.hipDeviceSynchronize()
            L = log()
            quiet(); pairs_up(L)
            expect = np.stack([2.0 * x[g * HOP + 3 * np.arange(cols)] + 1.0 for g in range(total_frames)]).astype(np.float32) \
                if total_frames else np.zeros((0, cols), np.float32)
            assert np.array_equal(out.get().reshape(-1, cols), expect), (world, total_frames, P, root, two)
            S = sync_log() if HOST_HIP else []
            for r in range(world):
                f0, f1 = frames[r]
                pieces = [shard_piece(f0, f1, P, k) for k in range(P)]
                assert [(fb, fe) for fb, fe, _, _ in calls[r]] == [(a - f0, b - f0) for a, b in pieces if b > a]
                assert all(st == USER for _, _, st, _ in calls[r])
                # the rank's exchange list, in issue order: S0, S1, G0, S2, G1, .. - piece k's samples are posted before
                # its compute is queued, its rows after, and the next piece's samples are already on their way
                mine = L[spans[r][0]:spans[r][1]]
                def S_ops(k):
                    if r == root:
                        return [("send", q, b - a) for q in range(world) if q != root
                                for a, b in [piece_samples(*frames[q], P, k, N, HOP)] if b > a]
                    a, b = piece_samples(f0, f1, P, k, N, HOP)
                    return [("recv", root, b - a)] if b > a else []
                def G_ops(k):
                    if r == root:
                        return [("recv", q, (q1 - q0) * cols) for q in range(world) if q != root
                                for q0, q1 in [shard_piece(*frames[q], P, k)] if q1 > q0]
                    q0, q1 = pieces[k]
                    return [("send", root, (q1 - q0) * cols)] if q1 > q0 else []
                want_ops, kinds, at_compute = S_ops(0), ["S"] * len(S_ops(0)), {}
                for k in range(P):
                    if k + 1 < P:
                        want_ops += S_ops(k + 1); kinds += ["S"] * len(S_ops(k + 1))
                    at_compute[k] = spans[r][0] + len(want_ops)
                    want_ops += G_ops(k); kinds += ["G"] * len(G_ops(k))
                assert [(e["op"], e["peer"], e["count"]) for e in mine] == want_ops, (world, r, P, root)
                got_at = [c[3] for c in calls[r]]
                assert got_at == [at_compute[k] for k in range(P) if pieces[k][1] > pieces[k][0]]
                sin = {e["stream"] for e, kd in zip(mine, kinds) if kd == "S"}
                sout = {e["stream"] for e, kd in zip(mine, kinds) if kd == "G"}
                assert len(sin) <= 1 and len(sout) <= 1 and USER not in (sin | sout) or not HOST_HIP
                if HOST_HIP and sin and sout:
                    assert (sin != sout) == two                      # two communicators: two exchange streams
                if HOST_HIP:
                    # the wiring of the three streams, call by call
                    seq = S[spans[r][2]:spans[r][3]]
                    A = seq[1][1]                                    # the scatter stream: first to wait for the fork event
                    B = seq[2][1] if two else A
                    e0 = seq[0][2]
                    assert seq[0] == ("rec", USER, e0) and seq[1] == ("wait", A, e0) and (not two or seq[2] == ("wait", B, e0))
                    assert A != USER and B != USER and (A != B) == two
                    body = seq[(3 if two else 2):]
                    evs = []
                    def take(kind, stream):
                        entry = body[take.i]
                        assert entry[:2] == (kind, stream), (entry, kind, stream, r, P)
                        take.i += 1
                        return entry[2]
                    take.i = 0
                    sev = {0: take("rec", A)}
                    for k in range(P):
                        if k + 1 < P:
                            sev[k + 1] = take("rec", A)
                        assert take("wait", USER) == sev[k]          # compute k behind its samples
                        cev = take("rec", USER)
                        assert take("wait", B) == cev                # gather k behind its rows
                        evs.append(cev)
                    j = take("rec", B)
                    assert take("wait", USER) == j                   # the caller's stream continues behind the last gather
                    if two:
                        assert take("rec", A) == e0 and take("wait", USER) == e0
                    assert take.i == len(body)
                    assert len(set(sev.values()) | set(evs) | {j, e0}) == 2 * P + 2
            for b in local + rows + [out]:
                b.free()
            checked += 1
        # (D) a compute callback that FAILS on one rank (its second non-empty piece): that rank still posts every exchange -
        # its peers' matching receives would otherwise wait in RCCL for ever -, the call returns the callback's code with
        # `stream` joined, the pieces of the other ranks arrive as they should
        bad = order[-1]
        if len([1 for k in range(P) if shard_piece(*frames[bad], P, k)[1] > shard_piece(*frames[bad], P, k)[0]]) >= 2:
            stub.aa_stub_reset()
            local = [Dev(s1 - s0 + 3) for s0, s1 in ranges]
            rows = [Dev((f1 - f0) * cols, np.zeros((f1 - f0) * cols, np.float32)) for f0, f1 in frames]
            out = Dev(total_frames * cols, np.zeros(total_frames * cols, np.float32))
            seen = {r: 0 for r in range(world)}
            rcs, counts = {}, {}
            for r in order:
                def compute(user, fb, fe, stream, r=r):
                    hip.hipDeviceSynchronize()
                    seen[r] += 1
                    if r == bad and seen[r] == 2:
                        return -5
                    assert not (r == bad and seen[r] > 2), "compute called again behind its own failure"
                    loc = local[r].get()
                    part = np.stack([2.0 * loc[i * HOP + 3 * np.arange(cols)] + 1.0 for i in range(fb, fe)]).astype(np.float32)
                    assert hip.hipMemcpy(rows[r].ptr + 4 * fb * cols, part.ctypes.data, 4 * part.size, 1) == 0
                    return 0
                n0 = stub.aa_stub_log_count()
                rcs[r] = lib.aa_sharded_pipeline(comms[r], comms[r], full.ptr if r == root else None, n_samples, N, HOP, cols, root,
                                                 local[r].ptr, local[r].n, rows[r].ptr if rows[r].n else None,
                                                 out.ptr if r == root and out.n else None, P, PIECE_FN(compute), None, None)
                counts[r] = stub.aa_stub_log_count() - n0
            hip.hipDeviceSynchronize()
            L = log()
            quiet(); pairs_up(L)                                     # every send has met its receive: nobody is left waiting
            assert rcs[bad] == -5 and all(rc == 0 for r, rc in rcs.items() if r != bad), rcs
            expect = np.stack([2.0 * x[g * HOP + 3 * np.arange(cols)] + 1.0 for g in range(total_frames)]).astype(np.float32)
            got = out.get().reshape(-1, cols)
            for r in range(world):
                f0, f1 = frames[r]
                if r != bad:
                    assert np.array_equal(got[f0:f1], expect[f0:f1]), (world, r)
                else:
                    q0, q1 = [shard_piece(f0, f1, P, k) for k in range(P) if shard_piece(f0, f1, P, k)[1] > shard_piece(f0, f1, P, k)[0]][0]
                    assert np.array_equal(got[q0:q1], expect[q0:q1])     # (the piece in front of the failure)
            for b in local + rows + [out]:
                b.free()
            checked += 1
        full.free()
    # argument checks: a piece index out of range, communicators of different ranks
    assert lib.aa_scatter_samples_piece(comms[0], None, 0, N, HOP, 0, None, 0, 4, 4, None) == -1
    assert lib.aa_gather_rows_piece(comms[0], None, 0, cols, 0, None, 0, 0, None) == -1
    assert lib.aa_sharded_pipeline(comms[0], comms2[1], None, 0, N, HOP, cols, 0, None, 0, None, None, 4,
                                   PIECE_FN(lambda *a: 0), None, None) == -1
    assert stub.aa_stub_group_depth() == 0
    if HOST_HIP:
        h = ctypes.CDLL(None)
        assert h.aa_hip_host_current_device() == 0, "a collective left another device selected"
    for c in comms + comms2:
        ok(lib.aa_comm_destroy(c))
print("scenarios", checked)
'''


def build_stub(name, tmp, extra=()):
    so = os.path.join(tmp, name + ".so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-Wall", "-o", so, os.path.join(STUBS, name + ".c"), "-ldl",
                           *extra])
    return so


def run_child(tmp_path, host_hip):
    so = os.path.join(ROOT, "audio_analyzer_rs_amd", "libaa_stft.so")
    assert os.path.exists(so), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    rccl = build_stub("rccl_loopback", str(tmp_path))
    env = dict(os.environ, AA_RCCL_PATH=rccl)
    if host_hip:
        env["LD_PRELOAD"] = build_stub("hip_host", str(tmp_path))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, rccl, "1" if host_hip else "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "scenarios" in r.stdout
    return int(r.stdout.split()[-1])


# per world: 4 whole-range cases x (2 + 2 x 3 sub-scenarios) + 6 piece cases x 2 pipelines, + (r5) the failing-callback
# pipeline in the piece cases that leave the failing rank two non-empty pieces (3 per world)
N_SCENARIOS = 3 * (4 * (2 + 2 * 3) + 6 * 2) + 9


def test_comm_call_lists_worlds_2_3_8_host(tmp_path):
    assert run_child(tmp_path, host_hip=True) == N_SCENARIOS


@pytest.mark.gpu
def test_comm_call_lists_worlds_2_3_8_device_buffers(tmp_path):
    assert run_child(tmp_path, host_hip=False) == N_SCENARIOS

"""The N > 1 branch of bench.py (BASELINE configs[3]: one corpus, frame-range sharded, strong
scaling; compute-only and scatter + compute + gather legs) executed by CI with world size 2 over
gloo.  `bench.run_sharded` is the very function the 8-GPU run calls; only the per-rank compute is
swapped for the CPU oracle (the HIP path needs a GPU) and the corpus is 3 s instead of 1 h."""
import json
import os
import socket
import sys

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


class OracleCompute:
    """CPU stand-in with the interface of bench.HipCompute."""

    cols, kernel, stream = 513, "cpu-oracle-standin", None

    def to_device(self, x_np):
        return torch.from_numpy(x_np)

    def empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32)

    def execute(self, x, nf, out):
        import oracle

        if nf > 0:
            out.copy_(torch.from_numpy(oracle.stft(x.numpy(), 1024, 256, kind="mag")))

    def execute_range(self, x, fb, fe, rows):
        import oracle

        if fe > fb:
            rows[fb:fe].copy_(torch.from_numpy(oracle.stft(x.numpy()[fb * 256:(fe - 1) * 256 + 1024], 1024, 256, kind="mag")))

    def sync(self):
        pass

    def events(self):
        return None


def _worker(rank, world, port, total, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        res = bench.run_sharded(dist, OracleCompute(), world, rank, total, steps=2, warmup=1, e2e_passes=2,
                                ramp_s=0.0)
        if rank == 0:
            json.dump(res, open(os.path.join(tmpdir, "result.json"), "w"))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharded_path_over_gloo(tmp_path, world):
    total = 3 * 48000
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    res = json.load(open(tmp_path / "result.json"))
    frames = (total - 1024) // 256 + 1
    assert res["n_gpus"] == world and res["n_ranks_seen"] == world
    assert res["scaling"] == "strong" and res["steps"] == 2 and res["warmup"] == 1
    assert res["config"]["frames_total"] == frames
    assert res["value"] > 0 and res["ms_per_step"] > 0
    assert abs(res["value"] - frames * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]
    # the rows rank 0 gathered from every rank agree with the float64 truth
    assert res["post_check_max_err"] <= res["post_check_tol"]
    for k in ("scatter_ms", "compute_ms", "gather_ms", "serial_total_ms", "total_ms", "frames_per_s"):
        assert res["e2e"][k] > 0
    # the overlapped leg delivered the same rows (its own post-check, piece edges included)
    assert res["e2e"]["overlap"]["pieces_per_rank"] == 8
    assert res["e2e"]["overlap"]["post_check_max_err"] <= res["post_check_tol"]
    # the overlap itself, by counting instead of by time (host-staged gloo has no transport to hide): every compute but
    # the last was queued with the next piece's samples on their way, every one but the first with rows still travelling
    assert res["e2e"]["overlap"]["computes_with_scatter_in_flight"] == 7
    assert res["e2e"]["overlap"]["computes_with_gather_in_flight"] == 7
    assert res["roofline"]["frames_per_launch"] == frames // world


def _worker_failing_legs(rank, world, port, total, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import audio_analyzer_rs_amd as aa
        import bench

        def broken(*a, **k):
            raise RuntimeError("link down (test)")

        aa.gather_samples = broken                       # the first call of the end-to-end legs, on every rank
        ctx = {}
        res = bench.run_sharded(dist, OracleCompute(), world, rank, total, steps=2, warmup=1, e2e_passes=2, ramp_s=0.0,
                                ctx=ctx)
        if rank == 0:
            json.dump({"res": res, "partial": ctx["partial"]}, open(os.path.join(tmpdir, "result.json"), "w"))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_bench_line_survives_failing_e2e_legs(tmp_path):
    """The end-to-end legs are the first code of bench.py that moves bytes between GPUs: when they raise, the line still
    carries the measured compute-only figure, a post-check of rank 0's own rows and the error; and the line run_multi's
    watchdog would print if they never came back (ctx["partial"]) is the same line."""
    total = 3 * 48000
    mp.spawn(_worker_failing_legs, args=(2, _free_port(), total, str(tmp_path)), nprocs=2, join=True)
    got = json.load(open(tmp_path / "result.json"))
    res, partial = got["res"], got["partial"]
    assert res["value"] > 0 and res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert "link down" in res["e2e"]["error"]
    assert res["post_check_max_err"] == res["post_check_local_max_err"] <= res["post_check_tol"]
    assert partial["value"] == res["value"] and "error" in partial["e2e"]
    assert {k for k in res if k != "e2e"} == {k for k in partial if k != "e2e"}


def test_overlap_counts_tell_schedules_apart():
    """aa.overlap_counts on hand-made traces: the designed schedule, the serial one (rejected: not the issue order), and
    a schedule that waits for every gather at once (no rows in flight behind a compute)."""
    sys.path.insert(0, ROOT)
    import audio_analyzer_rs_amd as aa

    P = 4
    good = [("S", 0)]
    for k in range(P):
        good += ([("S", k + 1)] if k + 1 < P else []) + [("W", k), ("C", k), ("G", k)]
    good.append(("J",))
    assert aa.overlap_counts(good, P) == {"pieces": P, "scatter_ahead": P - 1, "gather_behind": P - 1}
    serial = []
    for k in range(P):
        serial += [("S", k), ("W", k), ("C", k), ("G", k)]
    serial.append(("J",))
    with pytest.raises(ValueError):
        aa.overlap_counts(serial, P)
    assert aa.overlap_counts([("S", 0), ("W", 0), ("C", 0), ("G", 0), ("J",)], 1) == {"pieces": 1, "scatter_ahead": 0, "gather_behind": 0}


def test_bench_refuses_mismatched_world(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit):
        bench.main()


def test_bench_helpers_on_cpu(monkeypatch):
    """bench.py pieces that need no GPU: the row picker covers both ends and the ragged tail, the
    checker flags a wrong row, and `--gpus N` without WORLD_SIZE re-launches itself under
    torch.distributed.run on 127.0.0.1 with the same arguments (nothing touches the GPU before that)."""
    sys.path.insert(0, ROOT)
    import bench
    import audio_analyzer_rs_amd as aa

    ids = bench.pick_rows(112497, 96)
    assert len(ids) == 96 and set(range(16)) <= set(ids) and set(range(112497 - 16, 112497)) <= set(ids)
    assert list(bench.pick_rows(5, 64)) == [0, 1, 2, 3, 4]
    x = aa.signals.synth_corpus(0, 1024 + 9 * 256)
    import oracle

    rows = oracle.stft(x, 1024, 256, kind="mag")
    assert bench.check_rows(x, rows, np.arange(10), 1024, 256, "mag") < bench.TOL_VS_TRUTH
    bad = rows.copy()
    bad[3] = rows[4]
    assert bench.check_rows(x, bad, np.arange(10), 1024, 256, "mag") > 1e-2
    cx = oracle.stft(x, 1024, 256, kind="complex")
    assert bench.check_rows(x, cx, np.arange(10), 1024, 256, "complex") < bench.TOL_VS_TRUTH
    calls = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "--master-addr" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

"""CPU tests of the drop-in boundary: libaa_stft.so loads, exports every symbol that
include/aa_stft.h declares, and the GPU-free entry points behave (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry

    entry.build_library()          # hipcc cross-compiles gfx950 without a GPU
    import audio_analyzer_rs_amd as aa

    return aa.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "aa_stft.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aa_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from audio_analyzer_rs_amd import _lib

    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/aa_stft.h but not exported"
    # and the ctypes signature table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_header_is_plain_c_and_structs_match_the_ctypes_mirrors(tmp_path):
    """include/aa_stft.h is a C header (the Rust / cgo / ctypes bindings read it as such): it compiles
    as C99 with no warnings, and the two structs that cross the boundary have the layout the Python
    mirror (and the #[repr(C)] structs of INTEGRATION.md) assume."""
    import subprocess

    from audio_analyzer_rs_amd import _lib

    src = tmp_path / "layout.c"
    src.write_text(
        '#include "aa_stft.h"\n#include <stddef.h>\n#include <stdio.h>\n'
        "int main(void) {\n"
        '  printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(aa_stft_config), sizeof(aa_tuner_output),\n'
        "         offsetof(aa_tuner_output, cents), offsetof(aa_tuner_output, notes), offsetof(aa_tuner_output, accuracies),\n"
        "         offsetof(aa_tuner_output, base_freq), offsetof(aa_tuner_output, beat_position));\n"
        "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-o", str(exe), str(src)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    T = _lib.TunerOutputStruct
    assert got == [ctypes.sizeof(_lib.StftConfig), ctypes.sizeof(T), T.cents.offset, T.notes.offset,
                   T.accuracies.offset, T.base_freq.offset, T.beat_position.offset]


def test_abi_version_and_error_string(lib):
    # 4 (r5): the gates' per-frame mask has three bits and `calibrated` three modes, the state a fifth word
    # (calibration_done); flags bits 6 / 7; the flux is bit-exact; aa_diag_*
    assert lib.aa_abi_version() == 4
    assert isinstance(lib.aa_last_error(), bytes)


def test_hann_window_bit_exact_vs_golden(lib, golden):
    import audio_analyzer_rs_amd as aa

    for n in (256, 1024, 2048):
        assert np.array_equal(aa.hann_window(n), golden[f"n{n}_window"])


def test_bad_arguments_return_status_not_abort(lib):
    from audio_analyzer_rs_amd import _lib

    plan = ctypes.c_void_p()
    assert lib.aa_fft_create(1023, ctypes.byref(plan)) == _lib.AA_ERR_LEN     # odd length
    assert b"1023" in lib.aa_last_error()
    assert lib.aa_fft_create(2, ctypes.byref(plan)) == _lib.AA_ERR_LEN
    assert lib.aa_fft_create(1 << 20, ctypes.byref(plan)) == _lib.AA_ERR_LEN
    assert lib.aa_fft_create(1024, None) == _lib.AA_ERR_ARG
    assert lib.aa_hann_window(0, None) == _lib.AA_ERR_ARG
    cfg = _lib.StftConfig(1024, 0, 0, 1, 48000, 0, 0, 0.0, 0.0, 0.0)
    assert lib.aa_stft_plan_create(ctypes.byref(cfg), None, ctypes.byref(plan)) == _lib.AA_ERR_ARG
    assert lib.aa_stft_execute_channels_device(None, None, 4096, 4096, 2, None, 0, None) == _lib.AA_ERR_ARG
    assert lib.aa_fft_destroy(None) == 0 and lib.aa_stft_plan_destroy(None) == 0


def test_shard_ranges_cover_and_are_exact(lib):
    import audio_analyzer_rs_amd as aa

    for total in (0, 1, 7, 112_497, 674_997):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                f0, f1 = aa.shard_frames(total, world, r)
                assert f0 == prev and f1 >= f0
                assert f0 == total * r // world and f1 == total * (r + 1) // world
                prev = f1
            assert prev == total
    # halo of n_fft - hop samples shared with the next rank
    f0, f1 = aa.shard_frames(674_997, 8, 2)
    s0, s1 = aa.shard_samples(f0, f1, 1024, 256)
    g0, _ = aa.shard_frames(674_997, 8, 3)
    t0, _ = aa.shard_samples(g0, g0 + 1, 1024, 256)
    assert s0 == f0 * 256 and s1 == (f1 - 1) * 256 + 1024
    assert s1 - t0 == 1024 - 256
    assert aa.shard_samples(5, 5, 1024, 256) == (5 * 256, 5 * 256)


def test_no_cpu_fallback_without_gpu(lib):
    """On a machine without a HIP device a plan cannot be created: the product path
    fails loudly instead of computing on the CPU."""
    import torch

    import audio_analyzer_rs_amd as aa

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aa.AaError) as ei:
        aa.FftProcessor(1024)
    assert ei.value.code == -3
    with pytest.raises(aa.AaError):
        aa.Stft(1024, 256)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "audio_analyzer_rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "stft_oracle" not in text, f


def test_synthetic_corpus_is_shard_consistent():
    from audio_analyzer_rs_amd import signals

    a = signals.synth_corpus(0, 20_000)
    b = np.concatenate([signals.synth_corpus(0, 7_777), signals.synth_corpus(7_777, 12_223)])
    assert np.array_equal(a, b)
    assert np.abs(a).max() <= 0.97
    # the LCG is the reference's (metronome.rs:56-57)
    s, seq = 12345, []
    for _ in range(5):
        s = (s * 1103515245 + 12345) & 0x7FFFFFFF
        seq.append(s)
    assert np.array_equal(signals.lcg_states(12345, 0, 5), np.array(seq, dtype=np.uint64))
    assert np.array_equal(signals.lcg_states(12345, 2, 3), np.array(seq[2:], dtype=np.uint64))


def test_comm_argument_errors_without_a_gpu(lib):
    """aa_comm_*: argument checks come before anything touches RCCL or a device."""
    from audio_analyzer_rs_amd import _lib

    comm = ctypes.c_void_p()
    uid = ctypes.create_string_buffer(128)
    assert lib.aa_comm_create(None, 2, 0, ctypes.byref(comm)) == _lib.AA_ERR_ARG
    assert lib.aa_comm_create(uid, 0, 0, ctypes.byref(comm)) == _lib.AA_ERR_ARG
    assert lib.aa_comm_create(uid, 2, 2, ctypes.byref(comm)) == _lib.AA_ERR_ARG
    assert lib.aa_comm_unique_id(None) == _lib.AA_ERR_ARG
    assert lib.aa_comm_destroy(None) == _lib.AA_OK
    assert lib.aa_comm_world(None) == 0 and lib.aa_comm_rank(None) == -1
    assert lib.aa_comm_barrier(None, None) == _lib.AA_ERR_ARG
    assert lib.aa_gather_rows(None, None, 10, 513, 0, None, None) == _lib.AA_ERR_ARG
    assert lib.aa_scatter_samples(None, None, 10, 1024, 256, 0, None, 0, None) == _lib.AA_ERR_ARG
    assert isinstance(lib.aa_comm_backend(), bytes)


def test_stream_create_rejects_bad_hops_before_touching_a_device(lib):
    """aa_stream_create: hop == 0 would emit one frame for ever, hop > n_fft would drive the ring's
    `available` count negative (a usize underflow in the reference, stft.rs:436-437) - both are
    AA_ERR_ARG, checked before any plan (and so any device) is involved."""
    from audio_analyzer_rs_amd import _lib

    for n_fft, hop in ((1024, 0), (1024, -5), (1024, 1025), (0, 0)):
        cfg = _lib.StftConfig(n_fft, hop, _lib.AA_WINDOW_HANN_PERIODIC, _lib.AA_OUT_MAG, 48000, 0, 0, 0.0, 0.0, 0.0)
        h = ctypes.c_void_p()
        assert lib.aa_stream_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == _lib.AA_ERR_ARG
        assert not h.value
        assert b"hop" in lib.aa_last_error()

"""Static facts of the compiled gfx950 wave kernels against the record in tests/golden/kernel_static.json
(tools/kernel_static.py wrote it for the build that tools/check_counted_waits.py showed bit-identical to the
full-wait build, profiles/r03_counted_waits_check.txt).

Why: the chunk-top waits are COUNTED - `s_waitcnt vmcnt(K)` leaves exactly the row stores a wave issued behind
its LDS-DMA in flight - so they are only as right as the number of global_store instructions the compiler emits
per chunk (a store split in two, or merged away, would let a wave read samples whose DMA has not landed).  The
record pins, per kernel instantiation: global_store and global_load_lds instruction counts, the set of vmcnt
values waited for, scratch bytes (a spill adds vector-memory operations and costs 10 % and more), and the
waves-per-SIMD step of the register count.  A compiler upgrade that changes any of them fails here: re-run
tools/check_counted_waits.sh + .py on a GPU box, then refresh the record."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def waves_per_simd(vgprs):
    alloc = -(-vgprs // 8) * 8
    return min(8, 512 // alloc)


@pytest.fixture(scope="module")
def fresh():
    import kernel_static

    return kernel_static.facts()


def test_counted_wait_inputs_match_the_verified_build(fresh):
    with open(os.path.join(ROOT, "tests", "golden", "kernel_static.json")) as f:
        golden = json.load(f)
    assert set(fresh["kernels"]) == set(golden["kernels"]), "kernel instantiations changed: refresh the record"
    bad = []
    for name, g in golden["kernels"].items():
        k = fresh["kernels"][name]
        for key in ("global_store", "global_load_lds", "vmcnt_waits", "scratch_bytes", "m0_other_uses"):
            if k[key] != g[key]:
                bad.append((name, key, g[key], k[key]))
        if waves_per_simd(k["vgprs"]) != waves_per_simd(g["vgprs"]):
            bad.append((name, "waves_per_simd", waves_per_simd(g["vgprs"]), waves_per_simd(k["vgprs"])))
    assert not bad, f"compiler {fresh['compiler']!r} vs recorded {golden['compiler']!r}: {bad[:8]}"


def test_no_wave_kernel_spills_and_occupancy_as_planned(fresh):
    for name, k in fresh["kernels"].items():
        assert k["scratch_bytes"] == 0, (name, k)
        assert k["m0_other_uses"] == 0, (name, k)        # m0 belongs to the hand-issued LDS-DMA alone
        if name.startswith("stft_wave1024_kernel<") and name.split("<")[1][0] in "0123":
            assert waves_per_simd(k["vgprs"]) == 4, (name, k)        # four workgroups of four waves per CU
        if name.startswith("stft_wave2048_kernel") or "_mel_kernel" in name:
            assert waves_per_simd(k["vgprs"]) >= 3, (name, k)        # three workgroups per CU (the mel kinds: LDS and registers)
    # the bench kernel's counted wait leaves its five row stores (4 dwordx4 + 1 dwordx2 per frame pair) in flight
    assert 5 in fresh["kernels"]["stft_wave1024_kernel<1, true>"]["vmcnt_waits"]
    # the band-blocked mel kernels (r4): log-mel 1024 / 2048 and the fused MFCC kernel are in the record
    for name in ("stft_wave1024_mel_kernel<7>", "stft_wave1024_mel_kernel<8>", "stft_wave2048_mel_kernel<7>"):
        assert name in fresh["kernels"], sorted(fresh["kernels"])

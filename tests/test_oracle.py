"""CPU tests: the C restatement (oracle/) against the float64 truth, the committed golden
vectors, and the integer framing / ring-buffer rules of the reference."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_GEOMETRIES, GOLDEN_SIGNALS, TOL_VS_TRUTH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hann_matches_reference_formula(oracle_mod, golden):
    # stft.rs:641-648: periodic (denominator n), all f32.
    for n in (256, 1024, 2048):
        w = oracle_mod.hann_window(n)
        assert w.dtype == np.float32 and w.shape == (n,)
        assert w[0] == 0.0
        assert abs(w[n // 2] - 1.0) < 1e-6
        # periodic, not symmetric: w[1] == w[n-1], w[n-1] != 0
        assert abs(w[1] - w[n - 1]) < 1e-6 and w[n - 1] > 0
        w64 = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)
        assert np.max(np.abs(w - w64)) < 3e-7
        # numpy's own f32 evaluation differs by at most a few ulp from libm cosf
        assert np.max(np.abs(w - oracle_mod.truth.hann_f32(n))) < 2.5e-7
        assert np.array_equal(w, golden[f"n{n}_window"])


@pytest.mark.parametrize("n", [4, 8, 16, 64, 256, 512, 1024, 2048, 4096, 8192, 6, 10, 12, 30, 100])
def test_rfft_forward_vs_numpy_f64(oracle_mod, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    X = oracle_mod.rfft_forward(x)
    T = np.fft.rfft(x.astype(np.float64))
    assert X.shape == (n // 2 + 1,)
    assert np.max(np.abs(X - T)) / np.max(np.abs(T)) < 1e-6
    # fft.rs contract: X[0] and X[n/2] are purely real
    assert X[0].imag == 0.0 and X[n // 2].imag == 0.0


@pytest.mark.parametrize("n", [4, 16, 256, 1024, 2048, 6, 30])
def test_rfft_inverse_unnormalised(oracle_mod, n):
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal(n).astype(np.float32)
    X = oracle_mod.rfft_forward(x)
    y = oracle_mod.rfft_inverse(X, n)
    # realfft convention: inverse(forward(x)) = n * x
    assert np.max(np.abs(y / n - x)) < 5e-6


@pytest.mark.parametrize("n,hop", GOLDEN_GEOMETRIES)
@pytest.mark.parametrize("name", GOLDEN_SIGNALS)
def test_oracle_vs_golden(oracle_mod, golden, n, hop, name):
    x = golden[f"n{n}_{name}_x"]
    T = golden[f"n{n}_{name}_X"]
    assert T.shape == (oracle_mod.num_frames(len(x), n, hop), n // 2 + 1)
    C = oracle_mod.stft(x, n, hop, kind="complex")
    M = oracle_mod.stft(x, n, hop, kind="mag")
    err = oracle_mod.truth.max_normalised_error
    if name == "silence":
        assert not C.any() and not M.any()
        return
    assert err(C, T).max() < TOL_VS_TRUTH
    assert err(M, np.abs(T)).max() < TOL_VS_TRUTH


def test_num_frames_rule(oracle_mod):
    # stft.rs:273, 436-437: while available >= N { ...; available -= hop }
    nf = oracle_mod.num_frames
    assert nf(1023, 1024, 256) == 0
    assert nf(1024, 1024, 256) == 1
    assert nf(1279, 1024, 256) == 1
    assert nf(1280, 1024, 256) == 2
    # BASELINE configs
    assert nf(28_800_000, 1024, 256) == 112_497
    assert nf(172_800_000, 1024, 256) == 674_997
    assert nf(28_800_000, 2048, 512) == 56_247
    for L, n, h in [(5000, 256, 64), (4097, 2048, 512), (10_000, 1024, 100)]:
        cnt, avail = 0, L
        while avail >= n:
            cnt += 1
            avail -= h
        assert nf(L, n, h) == cnt


def test_stream_gap_free_equals_batch_framing(oracle_mod):
    # 1024-sample slots (mod.rs:126-128) through the tuner ring (8192, stft.rs:171)
    n, hop, ring = 2048, 512, 8192
    rng = np.random.default_rng(3)
    x = rng.standard_normal(1024 * 40).astype(np.float32)
    starts, frames = oracle_mod.stream_frame_starts([1024] * 40, n, hop, ring, slot_data=x,
                                                    return_frames=True)
    nf = oracle_mod.num_frames(len(x), n, hop)
    assert np.array_equal(starts, np.arange(nf) * hop)          # bit-exact frame indexing
    for t in (0, 1, nf // 2, nf - 1):
        assert np.array_equal(frames[t], x[t * hop:t * hop + n])


def test_stream_overflow_drops_oldest(oracle_mod):
    # stft.rs:262-266: more than ring_len pending -> read position jumps forward
    n, hop, ring = 256, 64, 4096                                 # onset geometry (onset.rs:122-124)
    starts = oracle_mod.stream_frame_starts([1024, 6000, 1024], n, hop, ring)
    # after the first push: frames at 0,64,...; 1024 samples -> (1024-256)/64+1 = 13 frames
    assert np.array_equal(starts[:13], np.arange(13) * 64)
    # second push: available = 1024-13*64 + 6000 = 6192 > 4096 -> drop 2096
    first_after = 13 * 64 + 2096
    assert starts[13] == first_after
    assert np.all(np.diff(starts[13:]) == 64)


def test_onset_flux_restatement(oracle_mod):
    # onset.rs:261-291 on a hand-checkable case: 2 frames x 5 bins, prev = 0
    m = np.array([[1, 2, 3, 4, 5], [5, 4, 3, 2, 1]], dtype=np.float32)
    flux, energy = oracle_mod.onset_flux(m)
    assert np.allclose(energy, [15, 15])
    w = 1 - np.arange(5) / 5
    sm0 = np.array([1, 2, 3, 4, 5.0])          # smoothing of a linear ramp is the ramp
    assert np.isclose(flux[0], np.sum(sm0 * w))
    d1 = np.array([5, 4, 3, 2, 1.0]) - m[0]
    assert np.isclose(flux[1], np.sum(np.maximum(d1, 0) * w))
    # carrying the previous row in == computing the two shards as one
    f2, e2 = oracle_mod.onset_flux(m[1:], prev0=m[0])
    assert f2[0] == flux[1] and e2[0] == energy[1]


def _burst_signal(seconds=4.0, sr=48000, seed=5):
    """Quiet noise with decaying tone bursts every 250 ms (what an onset detector listens for)."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    x = (2e-3 * rng.standard_normal(n)).astype(np.float32)
    t = np.arange(int(0.12 * sr)) / sr
    for k, start in enumerate(range(sr // 5, n - len(t), sr // 4)):
        f = 220.0 * (1.0 + 0.37 * k)
        x[start:start + len(t)] += (0.5 * np.exp(-t / 0.03) * np.sin(2 * np.pi * f * t)).astype(np.float32)
    return x


def test_onset_gates_restatement(oracle_mod):
    # onset.rs:298-371, 531-535 on the onset geometry: bursts fire, silence does not, state carries
    x = _burst_signal()
    mags = oracle_mod.stft(x, 256, 64, kind="mag")
    flux, energy = oracle_mod.onset_flux(mags)
    gf = np.float32(10.0 ** (-60.0 / 20.0) * 129 / 2.0)
    out, st, init = oracle_mod.onset_gates(mags, flux, energy, gf)
    fired = np.nonzero(out["flags"] & 16)[0]
    starts = np.arange(48000 // 5, len(x) - int(0.12 * 48000), 48000 // 4)
    assert init and len(fired) == len(starts)
    # every burst start is reported within one window of its frame
    assert np.all(np.abs(fired * 64 - (starts - 128)) <= 256)
    assert out["burst_count"].dtype == np.int32 and out["burst_count"].max() >= 3
    assert np.all(out["flux"][out["burst_count"] < 2] == 0.0)          # silence gate
    assert np.all((out["velocity"] >= 0.0) & (out["velocity"] <= 1.0))
    # two pieces with the carried state == one piece
    cut = 1111
    o1, s1, i1 = oracle_mod.onset_gates(mags[:cut], flux[:cut], energy[:cut], gf)
    o2, s2, _ = oracle_mod.onset_gates(mags[cut:], flux[cut:], energy[cut:], gf, state=s1, initialized=i1)
    for key in out:
        assert np.array_equal(np.concatenate([o1[key], o2[key]]), out[key]), key
    assert np.array_equal(s2, st)
    # the transport's inputs (onset.rs:393-396, 404-440).  An all-zero tick mask and calibrated = 1 are the call above.
    z, _, _ = oracle_mod.onset_gates(mags, flux, energy, gf, suppress=np.zeros(len(flux), np.uint8), calibrated=True)
    for key in out:
        assert np.array_equal(z[key], out[key]), key
    # a tick guard over every second fired onset: those are detected but dropped (bit 5), and - exactly as in the
    # reference - frames_since_onset is NOT reset by a dropped onset unless it is below 3 (:531-535), so the onset's
    # later frames may fire instead: every frame that fired without the mask and sits under it must not fire now
    sup = np.zeros(len(flux), np.uint8)
    for f in fired[::2]:
        sup[max(0, f - 2):f + 3] = 1
    m, _, _ = oracle_mod.onset_gates(mags, flux, energy, gf, suppress=sup)
    mf = m["flags"]
    assert not np.any((mf & 16) & (sup.astype(bool) * 16))
    assert np.all((mf[fired[::2]] & 32) != 0) and np.all((mf[fired[1::2]] & 16) != 0)
    assert np.array_equal((mf & 32) != 0, ((mf & 4) != 0) & (sup != 0))
    # the parts of the decision in front of the tick test do not depend on it
    for key in ("flux", "energy_ema", "max_excess", "burst_count", "velocity"):
        assert np.array_equal(m[key], out[key]), key
    assert np.array_equal(mf & 15, out["flags"] & 15)
    # before the calibration click has been played (onset.rs:406-408): everything detected, nothing fired, and a gated
    # onset does NOT restart frames_since_onset - so the frames behind it may be gated too (bit 6 more often than bit 4)
    u, _, _ = oracle_mod.onset_gates(mags, flux, energy, gf, calibrated=False)
    assert not np.any(u["flags"] & (16 | 128)) and np.array_equal(u["flags"] & 15, out["flags"] & 15)
    assert np.array_equal(out["flags"] & 64, (out["flags"] & 16) << 2)         # calibrated: gated == event
    assert np.all((u["flags"][fired] & 64) != 0) and int((u["flags"] & 64 > 0).sum()) >= len(fired)


def test_onset_calibration_modes(oracle_mod):
    """The stream that spans its own calibration (onset.rs:361-371, 404-440, 535-539): mode 2 carries
    calibration_done in the state; the gated onset that completes the calibration is consumed (bit 7, no event) and
    restarts frames_since_onset exactly as a fired one does."""
    x = _burst_signal()
    mags = oracle_mod.stft(x, 256, 64, kind="mag")
    nf = mags.shape[0]
    flux, energy = oracle_mod.onset_flux(mags)
    gf = np.float32(10.0 ** (-60.0 / 20.0) * 129 / 2.0)
    cal, st_cal, _ = oracle_mod.onset_gates(mags, flux, energy, gf)
    unc, _, _ = oracle_mod.onset_gates(mags, flux, energy, gf, calibrated=0)
    fired = np.nonzero(cal["flags"] & 16)[0]
    # (i) no mask: every frame is open - the first gated onset is consumed, the rest is the calibrated stream
    m2, st2, _ = oracle_mod.onset_gates(mags, flux, energy, gf, calibrated=2)
    want = cal["flags"].copy()
    want[fired[0]] = (int(want[fired[0]]) & 0xEF) | 128
    assert np.array_equal(m2["flags"], want)
    assert st2[129 + 3].view(np.int32) == 1 and st_cal[129 + 3].view(np.int32) == 1
    assert np.array_equal(st2[:129 + 3].view(np.uint32), st_cal[:129 + 3].view(np.uint32))
    # (ii) the click is played at frame `t_open` and residuals are plausible for 0.5 s (375 frames) behind it: before
    # that the stream is the uncalibrated one (gated onsets ignored, the counter runs on), the first gated onset
    # inside the window is consumed, from there on onsets are events
    t_open = int(fired[2]) - 40
    mask = np.zeros(nf, np.uint8)
    mask[t_open:t_open + 375] = 2
    w, st_w, _ = oracle_mod.onset_gates(mags, flux, energy, gf, suppress=mask, calibrated=2)
    used = np.nonzero(w["flags"] & 128)[0]
    assert len(used) == 1 and t_open <= used[0] < t_open + 375
    assert np.array_equal(w["flags"][:used[0]], unc["flags"][:used[0]])
    assert not np.any(w["flags"][:used[0]] & 16) and np.any(w["flags"][used[0] + 1:] & 16)
    assert np.all((w["flags"] & 64) >= ((w["flags"] & 16) << 2)) and (w["flags"][used[0]] & (64 | 16)) == 64
    # ... and in two calls with the state carried, cut in front of and behind the consuming onset
    for cut in (int(used[0]) - 5, int(used[0]) + 1, int(used[0]) + 400):
        a, sa, ia = oracle_mod.onset_gates(mags[:cut], flux[:cut], energy[:cut], gf, suppress=mask[:cut], calibrated=2)
        b, sb, _ = oracle_mod.onset_gates(mags[cut:], flux[cut:], energy[cut:], gf, suppress=mask[cut:], calibrated=2,
                                          state=sa, initialized=ia)
        for key in w:
            assert np.array_equal(np.concatenate([a[key], b[key]]), w[key]), (cut, key)
        assert np.array_equal(sb.view(np.uint32), st_w.view(np.uint32))
    # (iii) a window in which no onset is gated (implausible residuals everywhere else): the calibration stays open
    none = np.zeros(nf, np.uint8)
    none[5:8] = 2
    n, st_n, _ = oracle_mod.onset_gates(mags, flux, energy, gf, suppress=none, calibrated=2)
    assert np.array_equal(n["flags"], unc["flags"]) and st_n[129 + 3].view(np.int32) == 0
    # (iv) the timeout (onset.rs:361-371) completes it in front of the frame's own decision: the timed-out frame may fire
    tmo = np.zeros(nf, np.uint8)
    tmo[fired[3]:] = 4
    t, st_t, _ = oracle_mod.onset_gates(mags, flux, energy, gf, suppress=tmo, calibrated=2)
    assert np.array_equal(t["flags"][:fired[3]], unc["flags"][:fired[3]]) and not np.any(t["flags"] & 128)
    ev = np.nonzero(t["flags"] & 16)[0]
    assert len(ev) >= 1 and ev[0] >= fired[3] and st_t[129 + 3].view(np.int32) == 1
    # (v) the tick guard keeps its meaning under every mode (bit 0), and a tick frame is never gated
    both = mask.copy()
    both[used[0] - 2:used[0] + 3] |= 1
    k, _, _ = oracle_mod.onset_gates(mags, flux, energy, gf, suppress=both, calibrated=2)
    assert (k["flags"][used[0]] & (32 | 64 | 128)) == 32
    # (vi) no frames: the state comes back untouched and uninitialised
    z, st_z, init_z = oracle_mod.onset_gates(mags[:0], flux[:0], energy[:0], gf)
    assert not init_z and not st_z.any()


def test_div3_core_is_the_division(oracle_mod):
    """The flux kernel's x / 3 (csrc/aa_mel.hip flux_div3: multiply by RN(1/3), exact remainder by one FMA, one
    correcting FMA) over ALL 2^32 f32 bit patterns against the division of onset.rs:268: exactly +inf, -inf (their
    remainder is NaN) and -0 (the add loses its sign) differ - the three inputs the kernel hands to v_div_fixup_f32.
    The device runs the same sweep through the whole function (tests/test_gpu_parity.py::test_flux_div3_exhaustive)."""
    import threading

    parts = 16
    res = [None] * parts

    def run(k):
        res[k] = oracle_mod.div3_mismatches(k << 28, 1 << 28)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(parts)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    bad = sorted(b for n, bits in res for b in bits)
    assert sum(n for n, _ in res) == 3 and bad == [0x7F800000, 0x80000000, 0xFF800000]


def _chord_signal(seconds=3.0, sr=48000, seed=3):
    """Two sustained harmonic notes (A3, E4), then a third (C#5) joins; low noise."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n) / sr
    x = (1e-3 * rng.standard_normal(n)).astype(np.float32)
    for f0, amp, start in ((220.0, 0.3, 0.0), (329.63, 0.2, 0.0), (554.37, 0.25, seconds / 2)):
        gate = (t >= start).astype(np.float32)
        for h in range(1, 9):
            x += (gate * amp / h * np.sin(2 * np.pi * f0 * h * t)).astype(np.float32)
    return x


def test_extract_pitches_restatement(oracle_mod):
    # stft.rs:443-620 on the tuner geometry: the notes of a chord come out, strongest first,
    # harmonics are not reported as notes
    sr = 48000
    x = _chord_signal()
    mags = oracle_mod.stft(x, 2048, 512, kind="mag")
    nb = mags.shape[1]
    eff, _, _ = oracle_mod.tuner_floor(mags, np.float32(10.0 ** (-60.0 / 20.0) * nb / 2.0))
    freq, score, count = oracle_mod.extract_pitches(mags, eff, sr / 2048.0)
    early, late = slice(20, 120), slice(160, 270)
    assert np.all(count[early] == 2) and np.all(count[late] == 3)
    for sl, notes in ((early, (220.0, 329.63)), (late, (220.0, 329.63, 554.37))):
        got = np.sort(freq[sl, :len(notes)], axis=1)
        assert np.all(np.abs(got / np.array(notes, np.float32) - 1.0) < 5e-3)
    assert np.all(np.diff(score[late, :3], axis=1) <= 0.0)              # descending score
    assert np.all(freq[np.arange(8)[None, :] >= count[:, None]] == 0.0)  # unused slots stay 0
    # silence: nothing above the floor
    f0, s0, c0 = oracle_mod.extract_pitches(np.full((4, nb), 1e-4, np.float32), np.full((4, nb), 1e-2, np.float32),
                                            sr / 2048.0)
    assert np.all(c0 == 0) and np.all(f0 == 0.0)


def test_pitch_tracker_host_matches_restatement(oracle_mod):
    """PitchTracker (stft.rs:28-117) is sequential host logic on <= 8 raw pitches per frame: the
    product's C++ version (aa_pitch_tracker_*, no GPU involved) against the C restatement,
    bit-exact, with and without onsets; plus the hysteresis itself."""
    import audio_analyzer_rs_amd as aa

    sr = 48000
    x = _chord_signal()
    mags = oracle_mod.stft(x, 2048, 512, kind="mag")
    nb = mags.shape[1]
    eff, _, _ = oracle_mod.tuner_floor(mags, np.float32(10.0 ** (-60.0 / 20.0) * nb / 2.0))
    freq, score, count = oracle_mod.extract_pitches(mags, eff, sr / 2048.0)
    rng = np.random.default_rng(0)
    drop = rng.random(len(count)) < 0.15                   # missed detections exercise the life counter
    count = np.where(drop, 0, count).astype(np.int32)
    onset = (rng.random(len(count)) < 0.03).astype(np.uint8)
    for on in (None, onset):
        rf, rs, rc = oracle_mod.pitch_tracker(freq, score, count, on)
        trk = aa.PitchTracker()
        cut = 97                                           # two calls carry the tracks across
        f1, s1, c1 = trk.process(freq[:cut], score[:cut], count[:cut], None if on is None else on[:cut])
        f2, s2, c2 = trk.process(freq[cut:], score[cut:], count[cut:], None if on is None else on[cut:])
        assert np.array_equal(np.concatenate([c1, c2]), rc)
        assert np.array_equal(np.concatenate([f1, f2]), rf) and np.array_equal(np.concatenate([s1, s2]), rs)
    rf, rs, rc = oracle_mod.pitch_tracker(freq, score, count, None)
    assert rc[0] == 0 and rc.max() >= 3                     # a note needs two hits before it shows
    assert np.all(np.abs(rf[150:, 0][rc[150:] > 0] / 220.0 - 1.0) < 0.02)


def test_conditioner_host_matches_restatement(oracle_mod):
    """ "next" row f-3: the reducer thread's conditioning (mod.rs:351-478) and DynamicsTracker
    (dynamics.rs:152-360) - sample-sequential host logic in the reference and in the product
    (aa_conditioner_*, no GPU involved).  Product C++ vs the C restatement: bit-exact, also when fed
    in pieces; and the behaviour itself: gate, noise floor, AGC clamp, levels."""
    import audio_analyzer_rs_amd as aa

    rng = np.random.default_rng(2)
    x = np.concatenate([_burst_signal(seconds=12.0, seed=1), 0.5 * _chord_signal(seconds=8.0),
                        (3e-4 * rng.standard_normal(48000 * 12)).astype(np.float32)]).astype(np.float32)
    y, o = oracle_mod.condition(x)
    c = aa.Conditioner()
    y1, o1 = c.process(x[: 300 * 1024])
    y2, o2 = c.process(x[300 * 1024:])
    assert np.array_equal(np.concatenate([y1, y2]), y)
    for k in o:
        assert np.array_equal(np.concatenate([o1[k], o2[k]]), o[k]), k
    yp, op = oracle_mod.condition(x, pieces=[100, len(y) // 1024 - 100])
    assert np.array_equal(yp, y) and all(np.array_equal(op[k], o[k]) for k in o)
    assert np.abs(y).max() <= 0.97 + 1e-6                                   # peak-headroom clamp
    assert o["level"].min() == 0 and o["level"].max() >= 6                  # silence .. forte
    assert o["noise_floor_db"][0] == np.float32(-180.0)                     # empty history: 20 log10(1e-9)
    assert -80.0 < o["noise_floor_db"][-1] < -65.0                          # settles on the 3e-4 noise
    gf = aa.global_floor_per_frame(o["noise_floor_db"], 10, 2048, 512, 1025)
    assert gf.dtype == np.float32 and gf[0] == np.float32(10.0) ** (o["noise_floor_db"][1] / np.float32(20.0)) * np.float32(512.5)


def test_conditioner_long_session_wraps_both_histories(oracle_mod):
    """The play history holds 5000 playing slots (dynamics.rs:166-177) and the quiet history 256: a
    session long enough to wrap both (the product keeps the windows sorted incrementally, the
    restatement sorts them every slot as the reference does - same percentiles, bit for bit)."""
    import audio_analyzer_rs_amd as aa

    rng = np.random.default_rng(11)
    sr, n_play = 48000, 5600 * 1024
    t = np.arange(n_play, dtype=np.float64) / sr
    loud = 0.05 + 0.25 * (0.5 + 0.5 * np.sin(2 * np.pi * t / 7.3))          # level wanders over the session
    play = (loud * np.sin(2 * np.pi * 330.0 * t) + 0.3 * loud * np.sin(2 * np.pi * 660.0 * t)).astype(np.float32)
    quiet = (2e-4 * rng.standard_normal(300 * 1024)).astype(np.float32)
    x = np.concatenate([quiet, play, quiet, play[: 800 * 1024]])
    y, o = oracle_mod.condition(x)
    c = aa.Conditioner()
    cut = 2900 * 1024
    y1, o1 = c.process(x[:cut])
    y2, o2 = c.process(x[cut:])
    assert np.array_equal(np.concatenate([y1, y2]), y)
    for k in o:
        assert np.array_equal(np.concatenate([o1[k], o2[k]]), o[k]), k
    assert (o["level"] > 0).sum() > 5000 + 600                               # the play window wrapped
    assert (o["level"] == 0).sum() > 2 * 256                                 # and the quiet one


def test_tuner_floor_restatement(oracle_mod):
    # stft.rs:326-367 — first frame initialises, a steady loud bin freezes, quiet bins release
    nb = 4
    mags = np.tile(np.array([[10.0, 0.5, 0.02, 1.0]], np.float32), (50, 1))
    gf = 0.1
    eff, st, init = oracle_mod.tuner_floor(mags, gf)
    assert init and eff.shape == (50, nb)
    # frame 0: floor = max(mag, 5*gf) ; eff = min(floor, 2.5*gf)
    assert np.allclose(eff[0], np.minimum(np.maximum(mags[0], 0.5), 0.25))
    # constant input -> zero volatility; bin 0 is 20x above its floor of 10? no: floor starts AT mag
    assert np.allclose(st[1], mags[0])                      # prev_mag
    assert np.all(st[2] == 0.0)                             # volatility
    # feeding in two pieces with the carried state == one piece
    e1, s1, i1 = oracle_mod.tuner_floor(mags[:20] * np.linspace(1, 2, 20, dtype=np.float32)[:, None], gf)
    x = mags * 1.0
    x[:20] *= np.linspace(1, 2, 20, dtype=np.float32)[:, None]
    ea, sa, _ = oracle_mod.tuner_floor(x, gf)
    eb, sb, _ = oracle_mod.tuner_floor(x[20:], gf, state=s1, initialized=i1)
    assert np.array_equal(ea[20:], eb) and np.array_equal(sa, sb)


def test_oracle_header_says_parity_unpinned():
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "oracle", "stft_oracle.c")).read()
    assert "PARITY UNPINNED" in src and "TEST INFRASTRUCTURE ONLY" in src


def test_wav_format_matches_recorder_semantics(tmp_path):
    """ "next" row f-4 (input half): the recorder's WAV format (recorder.rs:60-85) and the decoder's
    scaling - integer work, exact."""
    import audio_analyzer_rs_amd as aa

    x = np.array([0.0, 1.0, -1.0, 2.0, -3.0, 0.5, -0.5, 1e-6, -1e-6, 0.999985, np.nan, 3.0517578e-05],
                 dtype=np.float32)
    q = aa.wav.encode_pcm16(x)
    assert q.dtype == np.int16
    assert q.tolist() == [0, 32767, -32767, 32767, -32767, 16383, -16383, 0, 0, 32766, -32767, 0]
    assert np.array_equal(aa.wav.decode_pcm16(q), q.astype(np.float32) / np.float32(32768.0))
    sig = aa.signals.synth_corpus(3, 48000)
    path = tmp_path / "rec.wav"
    aa.wav.write_wav(path, sig, 48000)
    back, sr = aa.wav.read_wav(path)
    assert sr == 48000 and back.dtype == np.float32 and back.shape == sig.shape
    assert np.array_equal(back, aa.signals.pcm16_roundtrip(sig))
    assert np.max(np.abs(back - sig)) <= 2.0 / 32768.0           # truncation + the 32767 / 32768 gain
    # stereo files come back as [frames, channels]
    import wave

    st = np.stack([aa.wav.encode_pcm16(sig[:1000]), aa.wav.encode_pcm16(-sig[:1000])], axis=1)
    with wave.open(str(tmp_path / "st.wav"), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(44100)
        w.writeframes(st.astype("<i2").tobytes())
    y, sr2 = aa.wav.read_wav(tmp_path / "st.wav")
    assert sr2 == 44100 and y.shape == (1000, 2) and np.array_equal(y[:, 0], aa.wav.decode_pcm16(st[:, 0]))
    # the onset JSON the front end polls (lib.rs:326-338)
    assert aa.wav.onsets_json([1.5, 2.25], [-96, 12], [0.5, 1.0]) == (
        '[{"beat_position":1.500000,"raw_sample_offset":-96,"velocity":0.5000},'
        '{"beat_position":2.250000,"raw_sample_offset":12,"velocity":1.0000}]')


def test_wav_reader_other_sample_formats(tmp_path):
    """f-4, the player's side: the reference plays files back through a general decoder (symphonia,
    src/generators/player.rs:171-232); read_wav takes the linear WAV sample formats such a decoder accepts -
    u8, s16, s24, s32, f32, f64, plain and WAVE_FORMAT_EXTENSIBLE headers, several channels, extra chunks, a
    truncated last frame - with the decoder's scale factors (parity unpinned: symphonia is not in the tree)."""
    import struct

    import audio_analyzer_rs_amd as aa

    def riff(fmt_body, payload, extra=b""):
        chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + extra + b"data" + struct.pack("<I", len(payload)) + payload
        return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks

    def fmt_chunk(tag, ch, sr, bits, extensible=False):
        align = ch * bits // 8
        if not extensible:
            return struct.pack("<HHIIHH", tag, ch, sr, sr * align, align, bits)
        guid = struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
        return struct.pack("<HHIIHHHHI", 0xFFFE, ch, sr, sr * align, align, bits, 22, bits, 3) + guid

    rng = np.random.default_rng(5)
    n, sr = 257, 22050
    cases = {
        "u8": (1, 8, rng.integers(0, 256, (n, 2)).astype(np.uint8), lambda q: (q.astype(np.float32) - 128.0) / 128.0),
        "s16": (1, 16, rng.integers(-32768, 32768, (n, 2)).astype("<i2"), lambda q: q.astype(np.float32) / 32768.0),
        "s32": (1, 32, rng.integers(-2**31, 2**31, (n, 2)).astype("<i4"), lambda q: (q.astype(np.float64) / 2.0**31).astype(np.float32)),
        "f32": (3, 32, rng.standard_normal((n, 2)).astype("<f4"), lambda q: q.astype(np.float32)),
        "f64": (3, 64, rng.standard_normal((n, 2)).astype("<f8"), lambda q: q.astype(np.float32)),
    }
    for name, (tag, bits, q, want) in cases.items():
        for ext in (False, True):
            path = tmp_path / f"{name}_{int(ext)}.wav"
            path.write_bytes(riff(fmt_chunk(tag, 2, sr, bits, ext), q.tobytes(), extra=b"LIST" + struct.pack("<I", 3) + b"abc\0"))
            y, got_sr = aa.wav.read_wav(path)
            assert got_sr == sr and y.dtype == np.float32 and y.shape == (n, 2), name
            assert np.array_equal(y, want(q).astype(np.float32)), name
    # 24-bit: three bytes per sample, sign extended; extremes and a mono file with a truncated last frame
    q24 = np.array([0, 1, -1, 8388607, -8388608, 123456, -654321], dtype=np.int64)
    b24 = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in q24)
    (tmp_path / "s24.wav").write_bytes(riff(fmt_chunk(1, 1, 48000, 24), b24 + b"\x01\x02"))
    y, got_sr = aa.wav.read_wav(tmp_path / "s24.wav")
    assert got_sr == 48000 and y.shape == (7,) and np.array_equal(y, (q24.astype(np.float64) / 8388608.0).astype(np.float32))
    assert y[3] < 1.0 and y[4] == -1.0
    # refusals
    (tmp_path / "alaw.wav").write_bytes(riff(fmt_chunk(6, 1, 8000, 8), b"\x00" * 16))
    (tmp_path / "bad.wav").write_bytes(b"RIFX" + b"\x00" * 40)
    (tmp_path / "nodata.wav").write_bytes(b"RIFF" + struct.pack("<I", 28) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt_chunk(1, 1, 8000, 16))
    # (a fmt chunk that says 0 bits / block align 0 must be refused as inconsistent, not divide by zero)
    (tmp_path / "zerobits.wav").write_bytes(riff(struct.pack("<HHIIHH", 1, 1, 8000, 0, 0, 0), b"\x00" * 16))
    for bad in ("alaw.wav", "bad.wav", "nodata.wav", "zerobits.wav"):
        with pytest.raises(ValueError):
            aa.wav.read_wav(tmp_path / bad)
    # 24-bit decoding runs in pieces of 2^22 samples: a file longer than one piece, every sample checked
    big = (np.arange(-(1 << 22) - 5, (1 << 22) + 6, dtype=np.int64) % (1 << 24)).astype(np.int64)
    big = np.where(big >= 1 << 23, big - (1 << 24), big).astype(np.int32)
    raw24 = (big.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]).tobytes()
    (tmp_path / "big24.wav").write_bytes(riff(fmt_chunk(1, 1, 48000, 24), raw24))
    y, _ = aa.wav.read_wav(tmp_path / "big24.wav")
    assert np.array_equal(y, (big.astype(np.float64) / 8388608.0).astype(np.float32))


def test_detector_restatements_vs_golden(oracle_mod):
    """tests/golden/detectors_golden.npz (make_detector_golden.py) pins the "next"-row restatements:
    integer / flag outputs exactly, float recurrences exactly (plain f32 arithmetic), the
    logf-based pitch values to 1e-6 (libm may differ by an ulp between builds)."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detectors_golden.npz"))
    mags = oracle_mod.stft(g["onset_x"], 256, 64, kind="mag")
    flux, energy = oracle_mod.onset_flux(mags)
    assert np.array_equal(flux, g["onset_flux"]) and np.array_equal(energy, g["onset_energy"])   # (sequential f32 sums)
    gates, state, _ = oracle_mod.onset_gates(mags, g["onset_flux"], g["onset_energy"],
                                            np.float32(10.0 ** (-60.0 / 20.0) * 129 / 2.0))
    for k in ("burst_count", "flags"):
        assert np.array_equal(gates[k], g["gate_" + k]), k
    for k in ("flux", "energy_ema", "max_excess", "velocity"):
        assert np.allclose(gates[k], g["gate_" + k], rtol=2e-6, atol=1e-7), k
    assert int((g["gate_flags"] & 16 > 0).sum()) == 3
    # the transport's inputs (tick guard mask, calibration flag): flags exactly
    masked, _, _ = oracle_mod.onset_gates(mags, g["onset_flux"], g["onset_energy"], np.float32(10.0 ** (-60.0 / 20.0) * 129 / 2.0),
                                          suppress=g["gate_suppress"])
    uncal, _, _ = oracle_mod.onset_gates(mags, g["onset_flux"], g["onset_energy"], np.float32(10.0 ** (-60.0 / 20.0) * 129 / 2.0),
                                         calibrated=False)
    assert np.array_equal(masked["flags"], g["gate_flags_masked"]) and np.array_equal(uncal["flags"], g["gate_flags_uncalibrated"])
    assert int((g["gate_flags_masked"] & 32 > 0).sum()) >= 1 and not np.any(g["gate_flags_uncalibrated"] & 16)
    # r5: a stream across its calibration click (mode 2: click played at a frame, plausible for 375 frames) and one that
    # runs into the calibration timeout - flags and the carried calibration_done exactly
    gf0 = np.float32(10.0 ** (-60.0 / 20.0) * 129 / 2.0)
    for tag in ("cal", "tmo"):
        got, st, _ = oracle_mod.onset_gates(mags, g["onset_flux"], g["onset_energy"], gf0, suppress=g["gate_mask_" + tag], calibrated=2)
        assert np.array_equal(got["flags"], g["gate_flags_" + tag]), tag
        assert np.array_equal(st.view(np.uint32), g["gate_state_" + tag].view(np.uint32)), tag
    assert int((g["gate_flags_cal"] & 128 > 0).sum()) == 1 and int((g["gate_flags_cal"] & 16 > 0).sum()) >= 1
    assert not np.any(g["gate_flags_tmo"] & 128) and int((g["gate_flags_tmo"] & 16 > 0).sum()) >= 1
    tm = oracle_mod.stft(g["tuner_x"], 2048, 512, kind="mag")
    nb = tm.shape[1]
    eff, tstate, _ = oracle_mod.tuner_floor(tm, np.float32(10.0 ** (-60.0 / 20.0) * nb / 2.0))
    assert np.allclose(eff[[0, 1, 10, eff.shape[0] - 1]], g["tuner_eff_rows"], rtol=2e-6)
    assert np.allclose(eff.astype(np.float64).sum(axis=1), g["tuner_eff_sum"], rtol=1e-6)
    freq, score, count = oracle_mod.extract_pitches(tm, eff, 48000.0 / 2048.0)
    assert np.array_equal(count, g["tuner_count"])
    assert np.allclose(freq, g["tuner_freq"], rtol=1e-6) and np.allclose(score, g["tuner_score"], rtol=1e-6)
    sf, ss, sc = oracle_mod.pitch_tracker(g["tuner_freq"], g["tuner_score"], g["tuner_count"])
    assert np.array_equal(sc, g["tracker_count"]) and np.array_equal(sf[:, :8], g["tracker_freq"])


def test_mel_definition_matches_committed_golden(oracle_mod):
    """The float64 log-mel / MFCC definition (oracle/truth.py; build-defined, parity unpinned by the
    reference) reproduces tests/golden/mel_golden.npz: a change of the mel scale, the normalisation,
    the log floor or the DCT convention shows up here before it shows up on the GPU."""
    path = os.path.join(ROOT, "tests", "golden", "mel_golden.npz")
    tr = oracle_mod.truth
    dct = tr.dct2_ortho(40, 128)
    with np.load(path, allow_pickle=False) as z:
        keys = sorted(k[:-2] for k in z.files if k.endswith("_x"))
        assert len(keys) == 6
        for key in keys:
            n = int(key[1:].split("_")[0])
            x, mel, mfcc = z[key + "_x"], z[key + "_mel"], z[key + "_mfcc"]
            P = np.abs(tr.stft_complex(x, n, n // 4, oracle_mod.hann_window(n))) ** 2
            got = P @ tr.mel_filterbank(48000, n, 128).T
            assert got.shape == mel.shape == (20, 128) and np.allclose(got, mel, rtol=1e-12, atol=0.0)
            got_mfcc = (10.0 * np.log10(np.maximum(got, 1e-10))) @ dct.T
            assert np.allclose(got_mfcc, mfcc, rtol=1e-10, atol=1e-10)

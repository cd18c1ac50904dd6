"""Generates tests/golden/detectors_golden.npz — small vectors for the "next"-row stages (onset
flux / gates, tuner floor / pitches / tracker).

Like stft_golden.npz these come from the build's own CPU restatement (the reference has no
fixtures and cannot run here): they pin the restatement against a committed, reviewable set of
numbers so that a later edit of oracle/stft_oracle.c cannot silently change its behaviour.

Run from the repo root:  python tests/golden/make_detector_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from test_oracle import _burst_signal, _chord_signal  # noqa: E402


def main():
    out = {}
    # onset geometry (onset.rs:122-125): 0.9 s of bursts
    x = _burst_signal(seconds=0.9, seed=77)
    mags = oracle.stft(x, 256, 64, kind="mag")
    flux, energy = oracle.onset_flux(mags)
    gf = np.float32(10.0 ** (-60.0 / 20.0) * 129 / 2.0)
    gates, state, _ = oracle.onset_gates(mags, flux, energy, gf)
    out["onset_x"] = x
    out["onset_flux"], out["onset_energy"] = flux, energy
    for k, v in gates.items():
        out["gate_" + k] = v
    out["onset_state"] = state
    # the transport's inputs (r4: onset.rs:393-396, 404-440): a tick guard over the frames around the second fired onset,
    # and the same stream before the latency calibration has completed
    fired = np.nonzero(gates["flags"] & 16)[0]
    sup = np.zeros(len(flux), np.uint8)
    sup[max(0, fired[1] - 2):fired[1] + 3] = 1
    masked, _, _ = oracle.onset_gates(mags, flux, energy, gf, suppress=sup)
    uncal, _, _ = oracle.onset_gates(mags, flux, energy, gf, calibrated=False)
    out["gate_suppress"], out["gate_flags_masked"], out["gate_flags_uncalibrated"] = sup, masked["flags"], uncal["flags"]
    # r5 (onset.rs:361-371, 404-440, 535-539): the stream across its own calibration - the click played 30 frames in
    # front of the second onset, residuals plausible for 0.5 s = 375 frames; and a stream whose calibration times out
    # at the third onset.  Mode 2: calibration_done travels in the state.
    cal = np.zeros(len(flux), np.uint8)
    cal[fired[1] - 30:fired[1] - 30 + 375] = 2
    tmo = np.zeros(len(flux), np.uint8)
    tmo[fired[2]:] = 4
    for tag, mk in (("cal", cal), ("tmo", tmo)):
        res, st, _ = oracle.onset_gates(mags, flux, energy, gf, suppress=mk, calibrated=2)
        out["gate_mask_" + tag], out["gate_flags_" + tag], out["gate_state_" + tag] = mk, res["flags"], st
    # tuner geometry (stft.rs:169-174): 0.8 s of a chord
    y = _chord_signal(seconds=0.8, seed=78)
    tm = oracle.stft(y, 2048, 512, kind="mag")
    nb = tm.shape[1]
    eff, tstate, _ = oracle.tuner_floor(tm, np.float32(10.0 ** (-60.0 / 20.0) * nb / 2.0))
    freq, score, count = oracle.extract_pitches(tm, eff, 48000.0 / 2048.0)
    sf, ss, sc = oracle.pitch_tracker(freq, score, count)
    out["tuner_x"] = y
    out["tuner_eff_rows"] = eff[[0, 1, 10, eff.shape[0] - 1]]
    out["tuner_eff_sum"] = eff.astype(np.float64).sum(axis=1)
    out["tuner_state"] = tstate
    out["tuner_freq"], out["tuner_score"], out["tuner_count"] = freq, score, count
    out["tracker_freq"], out["tracker_score"], out["tracker_count"] = sf[:, :8], ss[:, :8], sc
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "detectors_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()

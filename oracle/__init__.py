"""CPU oracle for the STFT hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the timed CPU leg.  The
product package (``audio_analyzer_rs_amd``) never imports it.

PARITY UNPINNED by the reference: it holds no test, fixture or golden vector for
src/dsp/fft.rs, src/audio_io/stft.rs or src/analysis/onset.rs, and it is Rust,
which cannot be built in this image.  The oracle is pinned against a float64
``numpy.fft.rfft`` truth instead (``oracle.truth``).  The one stage the reference does pin is
the note / interval naming of TunerOutput (known-answer tests src/analysis/theory.rs:404-627):
``oracle/theory_oracle.c``, checked against those in ``tests/test_theory.py``.
"""
from .capi import (  # noqa: F401
    Oracle,
    build_oracle,
    condition,
    div3_mismatches,
    extract_pitches,
    interval,
    midi_from_freq,
    midi_to_freq,
    note_from_freq,
    note_to_freq,
    hann_window,
    num_frames,
    onset_flux,
    onset_gates,
    pitch_tracker,
    rfft_forward,
    rfft_inverse,
    stft,
    stream_frame_starts,
    tuner_floor,
    tuner_output,
)
from . import truth  # noqa: F401

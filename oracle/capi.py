"""ctypes bindings for oracle/libstft_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libstft_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build_oracle(force=False):
    """Compile oracle/stft_oracle.c + theory_oracle.c with the committed Makefile (gcc)."""
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("stft_oracle.c", "theory_oracle.c"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def _ptr(a):
    return a.ctypes.data_as(_f32p)


class Oracle:
    _inst = None

    def __init__(self):
        if not os.path.exists(_SO):
            build_oracle()
        lib = ctypes.CDLL(_SO)
        lib.aao_hann_window.argtypes = [ctypes.c_int, _f32p]
        lib.aao_hann_window.restype = None
        lib.aao_num_frames.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        lib.aao_num_frames.restype = ctypes.c_int64
        lib.aao_plan_create.argtypes = [ctypes.c_int]
        lib.aao_plan_create.restype = ctypes.c_void_p
        lib.aao_plan_destroy.argtypes = [ctypes.c_void_p]
        lib.aao_plan_destroy.restype = None
        lib.aao_rfft_forward.argtypes = [ctypes.c_void_p, _f32p, _f32p]
        lib.aao_rfft_forward.restype = ctypes.c_int
        lib.aao_rfft_inverse.argtypes = [ctypes.c_void_p, _f32p, _f32p]
        lib.aao_rfft_inverse.restype = ctypes.c_int
        lib.aao_magnitude.argtypes = [ctypes.c_int, _f32p, _f32p]
        lib.aao_magnitude.restype = None
        lib.aao_stft_range.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                       _f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, _f32p]
        lib.aao_stft_range.restype = ctypes.c_int64
        lib.aao_stream_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.aao_stream_create.restype = ctypes.c_void_p
        lib.aao_stream_destroy.argtypes = [ctypes.c_void_p]
        lib.aao_stream_destroy.restype = None
        lib.aao_stream_push.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, _i64p, _f32p,
                                        ctypes.c_int64]
        lib.aao_stream_push.restype = ctypes.c_int64
        lib.aao_onset_flux.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p, _f32p]
        lib.aao_onset_flux.restype = None
        lib.aao_div3_mismatches.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        lib.aao_div3_mismatches.restype = ctypes.c_uint64
        lib.aao_onset_gates.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p, _f32p,
                                        ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, _f32p,
                                        ctypes.POINTER(ctypes.c_int), _f32p, _f32p, _f32p,
                                        ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8), _f32p]
        lib.aao_onset_gates.restype = None
        lib.aao_extract_pitches.argtypes = [_f32p, _f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, _f32p, _f32p,
                                            ctypes.POINTER(ctypes.c_int32)]
        lib.aao_extract_pitches.restype = None
        lib.aao_extract_pitches_margin.argtypes = lib.aao_extract_pitches.argtypes + [_f32p]
        lib.aao_extract_pitches_margin.restype = None
        lib.aao_pitch_tracker.argtypes = [_f32p, _f32p, ctypes.POINTER(ctypes.c_int32),
                                          ctypes.POINTER(ctypes.c_uint8), ctypes.c_int64, _f32p,
                                          ctypes.POINTER(ctypes.c_int), ctypes.c_int, _f32p, _f32p,
                                          ctypes.POINTER(ctypes.c_int32)]
        lib.aao_pitch_tracker.restype = None
        lib.aao_conditioner_create.argtypes = [ctypes.c_float, ctypes.c_int]
        lib.aao_conditioner_create.restype = ctypes.c_void_p
        lib.aao_conditioner_destroy.argtypes = [ctypes.c_void_p]
        lib.aao_conditioner_destroy.restype = None
        lib.aao_conditioner_process.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int64, _f32p, _f32p, _f32p, _f32p,
                                                ctypes.POINTER(ctypes.c_int32)]
        lib.aao_conditioner_process.restype = None
        lib.aao_tuner_floor.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p,
                                        ctypes.POINTER(ctypes.c_int), _f32p]
        lib.aao_tuner_floor.restype = None
        lib.aao_lcg_next.argtypes = [ctypes.c_uint32]
        lib.aao_lcg_next.restype = ctypes.c_uint32
        # theory_oracle.c
        lib.aao_note_from_freq.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_char_p, _f32p]
        lib.aao_note_from_freq.restype = None
        lib.aao_note_to_freq.argtypes = [ctypes.c_char_p, ctypes.c_float, ctypes.c_float, _f32p]
        lib.aao_note_to_freq.restype = ctypes.c_int
        lib.aao_midi_from_freq.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_int32), _f32p]
        lib.aao_midi_from_freq.restype = None
        lib.aao_midi_to_freq.argtypes = [ctypes.c_int32, ctypes.c_float, ctypes.c_float]
        lib.aao_midi_to_freq.restype = ctypes.c_float
        lib.aao_interval.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, _f32p]
        lib.aao_interval.restype = None
        lib.aao_tuner_output.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                         ctypes.c_char_p, _f32p, ctypes.POINTER(ctypes.c_int32), ctypes.c_char_p,
                                         _f32p]
        lib.aao_tuner_output.restype = None
        self.lib = lib

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst


def hann_window(n):
    """Periodic f32 Hann (reference: src/audio_io/stft.rs:641-648)."""
    w = np.empty(n, dtype=np.float32)
    Oracle.get().lib.aao_hann_window(n, _ptr(w))
    return w


def num_frames(n_samples, n_fft, hop):
    return int(Oracle.get().lib.aao_num_frames(n_samples, n_fft, hop))


def rfft_forward(x):
    """Unnormalised R2C of one f32 frame -> complex64[n/2+1] (src/dsp/fft.rs:33-35)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    lib = Oracle.get().lib
    p = lib.aao_plan_create(n)
    if not p:
        raise ValueError(f"oracle: unsupported FFT length {n}")
    out = np.empty(2 * (n // 2 + 1), dtype=np.float32)
    try:
        rc = lib.aao_rfft_forward(p, _ptr(x), _ptr(out))
    finally:
        lib.aao_plan_destroy(p)
    if rc != 0:
        raise RuntimeError(f"oracle rfft_forward rc={rc}")
    return out.view(np.complex64)


def rfft_inverse(spec, n):
    """Unnormalised C2R (src/dsp/fft.rs:39-41): returns n*x for spec = rfft(x)."""
    spec = np.ascontiguousarray(spec, dtype=np.complex64)
    assert spec.shape[0] == n // 2 + 1
    lib = Oracle.get().lib
    p = lib.aao_plan_create(n)
    if not p:
        raise ValueError(f"oracle: unsupported FFT length {n}")
    out = np.empty(n, dtype=np.float32)
    try:
        rc = lib.aao_rfft_inverse(p, _ptr(spec.view(np.float32)), _ptr(out))
    finally:
        lib.aao_plan_destroy(p)
    if rc < 0:
        raise RuntimeError(f"oracle rfft_inverse rc={rc}")
    return out


def stft(x, n_fft, hop, kind="mag", window=None, f0=0, f1=None):
    """Frame loop of stft.rs:273-318 over a gap-free buffer.

    kind: "complex" -> complex64[frames, n/2+1]; "mag" -> float32[frames, n/2+1].
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    nf = num_frames(x.shape[0], n_fft, hop)
    if f1 is None:
        f1 = nf
    if window is None:
        window = hann_window(n_fft)
    window = np.ascontiguousarray(window, dtype=np.float32)
    nb = n_fft // 2 + 1
    k = 0 if kind == "complex" else 1
    out = np.empty((f1 - f0, nb * (2 if k == 0 else 1)), dtype=np.float32)
    if f1 - f0 == 0:
        return out.view(np.complex64) if k == 0 else out
    rc = Oracle.get().lib.aao_stft_range(_ptr(x), x.shape[0], n_fft, hop, _ptr(window), k,
                                         f0, f1, _ptr(out))
    if rc < 0:
        raise RuntimeError(f"oracle stft rc={rc}")
    return out.view(np.complex64) if k == 0 else out


def stream_frame_starts(slot_lens, n_fft, hop, ring_len, slot_data=None, return_frames=False):
    """Run the ring-buffer ingest (stft.rs:240-266, 273, 436-437) over a slot sequence.

    slot_lens: number of samples pushed before each drain (the reference pushes
    every queued slot and then drains; pass one combined length per drain).
    Returns the absolute first-sample index of every emitted frame (and the frames).
    """
    lib = Oracle.get().lib
    s = lib.aao_stream_create(n_fft, hop, ring_len)
    if not s:
        raise ValueError("bad stream params")
    starts, frames = [], []
    pos = 0
    try:
        for L in slot_lens:
            if slot_data is None:
                slot = np.zeros(L, dtype=np.float32)
            else:
                slot = np.ascontiguousarray(slot_data[pos:pos + L], dtype=np.float32)
            pos += L
            cap = L // hop + ring_len // hop + 2
            st = np.empty(cap, dtype=np.int64)
            fr = np.empty((cap, n_fft), dtype=np.float32) if return_frames else None
            n = lib.aao_stream_push(s, _ptr(slot), L, st.ctypes.data_as(_i64p),
                                    _ptr(fr) if fr is not None else None, cap)
            starts.extend(st[:n].tolist())
            if return_frames:
                frames.append(fr[:n].copy())
    finally:
        lib.aao_stream_destroy(s)
    if return_frames:
        return np.array(starts, dtype=np.int64), (np.concatenate(frames) if frames else
                                                  np.empty((0, n_fft), np.float32))
    return np.array(starts, dtype=np.int64)


def onset_flux(mags, prev0=None):
    """Spectral flux and frame energy (src/analysis/onset.rs:261-291), sequential f32."""
    mags = np.ascontiguousarray(mags, dtype=np.float32)
    nf, nb = mags.shape
    flux = np.empty(nf, dtype=np.float32)
    energy = np.empty(nf, dtype=np.float32)
    p0 = None if prev0 is None else np.ascontiguousarray(prev0, dtype=np.float32)
    Oracle.get().lib.aao_onset_flux(_ptr(mags), nf, nb, _ptr(p0) if p0 is not None else None,
                                    _ptr(flux), _ptr(energy))
    return flux, energy


def tuner_floor(mags, global_floor, state=None, initialized=False):
    """Per-bin adaptive noise floor of the tuner (src/audio_io/stft.rs:326-367).
    Returns (effective_floor [frames, bins], state [3, bins], initialized)."""
    mags = np.ascontiguousarray(mags, dtype=np.float32)
    nf, nb = mags.shape
    gf = np.ascontiguousarray(np.broadcast_to(np.asarray(global_floor, dtype=np.float32), (nf,)))
    st = np.zeros((3, nb), dtype=np.float32) if state is None else np.array(state, dtype=np.float32)
    init = ctypes.c_int(1 if initialized else 0)
    eff = np.empty((nf, nb), dtype=np.float32)
    Oracle.get().lib.aao_tuner_floor(_ptr(mags), nf, nb, _ptr(gf), _ptr(st), ctypes.byref(init), _ptr(eff))
    return eff, st, bool(init.value)


def div3_mismatches(first, count, cap=16):
    """Bit patterns in [first, first + count) for which the flux kernel's three-operation x / 3 core differs from the
    division (test support; releases the GIL, so ranges can run on threads).  Returns (count, first `cap` patterns)."""
    out = (ctypes.c_uint32 * cap)()
    n = Oracle.get().lib.aao_div3_mismatches(first, count, out, cap)
    return int(n), [int(v) for v in out[:min(int(n), cap)]]


def onset_gates(mags, flux, energy, global_floor, state=None, initialized=False, suppress=None, calibrated=True):
    """Onset detector logic after the flux (src/analysis/onset.rs:298-371, 531-535), sequential f32; suppress
    (uint8 per frame: bit 0 inside the metronome's tick guard, :393-396; bit 1 calibration click played and the residual
    plausible, :405-431; bit 2 calibration timed out, :361-371) and calibrated (False / 0: running, click not played;
    True / 1: done; 2: calibrating, completion carried in the state) are the transport's inputs.
    Returns (dict of per-frame arrays, state [bins + 4], initialized)."""
    mags = np.ascontiguousarray(mags, dtype=np.float32)
    nf, nb = mags.shape
    flux = np.ascontiguousarray(flux, dtype=np.float32)
    energy = np.ascontiguousarray(energy, dtype=np.float32)
    gf = np.ascontiguousarray(np.broadcast_to(np.asarray(global_floor, dtype=np.float32), (nf,)))
    st = np.zeros(nb + 4, dtype=np.float32) if state is None else np.array(state, dtype=np.float32)
    assert st.shape == (nb + 4,)
    init = ctypes.c_int(1 if initialized else 0)
    out = {"flux": np.empty(nf, np.float32), "energy_ema": np.empty(nf, np.float32),
           "max_excess": np.empty(nf, np.float32), "burst_count": np.empty(nf, np.int32),
           "flags": np.empty(nf, np.uint8), "velocity": np.empty(nf, np.float32)}
    Oracle.get().lib.aao_onset_gates(
        _ptr(mags), nf, nb, _ptr(flux), _ptr(energy), _ptr(gf),
        None if suppress is None else np.ascontiguousarray(suppress, dtype=np.uint8).ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
        int(calibrated), _ptr(st), ctypes.byref(init), _ptr(out["flux"]),
        _ptr(out["energy_ema"]), _ptr(out["max_excess"]),
        out["burst_count"].ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        out["flags"].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _ptr(out["velocity"]))
    return out, st, bool(init.value)


def extract_pitches(mags, eff_floor, bin_width, min_freq=24.0, max_freq=10000.0, margin=False):
    """Multi-pitch extraction of the tuner (src/audio_io/stft.rs:443-620) per frame, sequential f32.
    Returns (freq [frames, 8], score [frames, 8], count [frames]); with margin=True also, per frame, the smallest
    relative distance of any value the frame's decisions hang on to the boundary of its comparison (test support: a
    frame with a margin of ~1e-7 can come out differently under a logf that differs in its last bit)."""
    mags = np.ascontiguousarray(mags, dtype=np.float32)
    eff_floor = np.ascontiguousarray(eff_floor, dtype=np.float32)
    nf, nb = mags.shape
    assert eff_floor.shape == mags.shape
    freq = np.zeros((nf, 8), np.float32)
    score = np.zeros((nf, 8), np.float32)
    count = np.zeros(nf, np.int32)
    if margin:
        mg = np.zeros(nf, np.float32)
        Oracle.get().lib.aao_extract_pitches_margin(_ptr(mags), _ptr(eff_floor), nf, nb, bin_width, min_freq, max_freq,
                                                    _ptr(freq), _ptr(score),
                                                    count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _ptr(mg))
        return freq, score, count, mg
    Oracle.get().lib.aao_extract_pitches(_ptr(mags), _ptr(eff_floor), nf, nb, bin_width, min_freq, max_freq,
                                         _ptr(freq), _ptr(score),
                                         count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return freq, score, count


def pitch_tracker(freq, score, count, onset=None, cap=32):
    """PitchTracker::process (src/audio_io/stft.rs:28-117) over all frames from a fresh tracker.
    Returns (stable freq [frames, cap], score [frames, cap], count [frames])."""
    freq = np.ascontiguousarray(freq, dtype=np.float32)
    score = np.ascontiguousarray(score, dtype=np.float32)
    count = np.ascontiguousarray(count, dtype=np.int32)
    nf = freq.shape[0]
    on = None if onset is None else np.ascontiguousarray(onset, dtype=np.uint8)
    tracks = np.zeros((cap, 3), np.float32)
    n_tracks = ctypes.c_int(0)
    of = np.zeros((nf, cap), np.float32)
    os_ = np.zeros((nf, cap), np.float32)
    oc = np.zeros(nf, np.int32)
    Oracle.get().lib.aao_pitch_tracker(
        _ptr(freq), _ptr(score), count.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        None if on is None else on.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), nf, _ptr(tracks),
        ctypes.byref(n_tracks), cap, _ptr(of), _ptr(os_), oc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return of, os_, oc


def condition(samples, sample_rate=48000.0, slot_len=1024, pieces=None):
    """Reducer-thread conditioning (src/audio_io/mod.rs:351-478 + dynamics.rs:152-360) of whole slots.
    pieces: optional list of slot counts to feed in separate calls (state carried).
    Returns (conditioned samples, dict of per-slot arrays)."""
    x = np.array(samples, dtype=np.float32)
    n_slots = len(x) // slot_len
    x = x[: n_slots * slot_len].copy()
    out = {"noise_floor_db": np.empty(n_slots, np.float32), "rms_db": np.empty(n_slots, np.float32),
           "gain_db": np.empty(n_slots, np.float32), "session_median_db": np.empty(n_slots, np.float32),
           "level": np.empty(n_slots, np.int32)}
    lib = Oracle.get().lib
    c = lib.aao_conditioner_create(sample_rate, slot_len)
    try:
        done = 0
        for cnt in (pieces or [n_slots]):
            sl = slice(done, done + cnt)
            seg = x[done * slot_len:(done + cnt) * slot_len]
            args = [out[k][sl] for k in ("noise_floor_db", "rms_db", "gain_db", "session_median_db")]
            lv = out["level"][sl]
            lib.aao_conditioner_process(c, _ptr(seg), cnt, *[_ptr(a) for a in args],
                                        lv.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
            done += cnt
    finally:
        lib.aao_conditioner_destroy(c)
    return x, out


# ---- theory_oracle.c: note / interval naming and the TunerOutput assembly ----

def note_from_freq(freq, base_freq=440.0):
    """Note::from_freq (src/analysis/theory.rs:199-232) -> (name, cents as np.float32)."""
    name = ctypes.create_string_buffer(8)
    cents = ctypes.c_float()
    Oracle.get().lib.aao_note_from_freq(freq, base_freq, name, ctypes.byref(cents))
    return name.value.decode(), np.float32(cents.value)


def note_to_freq(name, base_freq=440.0, cents=0.0):
    """Note::try_new + to_freq (theory.rs:104-197); None when the name does not parse."""
    f = ctypes.c_float()
    rc = Oracle.get().lib.aao_note_to_freq(name.encode(), cents, base_freq, ctypes.byref(f))
    return None if rc else np.float32(f.value)


def midi_from_freq(freq, base_freq=440.0):
    midi = ctypes.c_int32()
    cents = ctypes.c_float()
    Oracle.get().lib.aao_midi_from_freq(freq, base_freq, ctypes.byref(midi), ctypes.byref(cents))
    return midi.value, np.float32(cents.value)


def midi_to_freq(midi, cents=0.0, base_freq=440.0):
    return np.float32(Oracle.get().lib.aao_midi_to_freq(midi, cents, base_freq))


def interval(freqs, system=0):
    """Interval::new (theory.rs:289-381) -> (name, accuracy)."""
    f = np.ascontiguousarray(freqs, dtype=np.float32)
    name = ctypes.create_string_buffer(8)
    acc = ctypes.c_float()
    Oracle.get().lib.aao_interval(_ptr(f) if len(f) else None, len(f), system, name, ctypes.byref(acc))
    return name.value.decode(), np.float32(acc.value)


def tuner_output(freqs, scores, mode=0, system=0, base_freq=440.0):
    """Body of Tuner::run for one non-empty pitch list (src/analysis/tuner.rs:153-193)."""
    f = np.ascontiguousarray(freqs, dtype=np.float32)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    label = ctypes.create_string_buffer(512)
    names = ctypes.create_string_buffer(8 * 32)
    acc = np.zeros(32, np.float32)
    cents = ctypes.c_float()
    n = ctypes.c_int32()
    Oracle.get().lib.aao_tuner_output(_ptr(f), _ptr(s), len(f), mode, system, base_freq, label, ctypes.byref(cents),
                                      ctypes.byref(n), names, _ptr(acc))
    raw = names.raw
    return {"label": label.value.decode(), "cents": np.float32(cents.value),
            "notes": [raw[8 * i:8 * i + 8].split(b"\0")[0].decode() for i in range(n.value)],
            "accuracies": acc[:n.value].copy()}

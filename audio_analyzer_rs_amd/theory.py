"""Host mirror of the reference's note / interval naming (src/analysis/theory.rs) and of the
tuner thread that turns stable pitch lists into `TunerOutput` (src/analysis/tuner.rs).

Every value comes from libaa_stft.so (aa_note_*, aa_midi_*, aa_interval, aa_tuner_output_*);
the classes only carry the reference's names and calling conventions so that the parity tests
read like the reference's own (theory.rs:404-627).
"""
import ctypes

import numpy as np

from . import _lib

MODES = {"MultiPitch": 0, "SinglePitch": 1}
SYSTEMS = {"EqualTemperament": 0, "JustIntonation": 1, "Pythagorean": 2}
NOTES_MAX = 32


def _system_code(system):
    if system is None:
        return 0
    return SYSTEMS[system] if isinstance(system, str) else int(system)


class Note:
    """`Note` (theory.rs:93-254): `Note.from_freq(f, base)`, `Note.try_new("C#4")`, `get_name()`,
    `get_cents()`, `to_freq(base)`."""

    def __init__(self, name, cents=0.0):
        self._name = name
        self._cents = np.float32(cents)

    @classmethod
    def from_freq(cls, freq, base_freq=None):
        name = ctypes.create_string_buffer(8)
        cents = ctypes.c_float()
        lib = _lib.load()
        _lib.check(lib.aa_note_from_freq(float(freq), 440.0 if base_freq is None else float(base_freq), name,
                                         ctypes.byref(cents)))
        return cls(name.value.decode(), cents.value)

    @classmethod
    def try_new(cls, name):
        """Parses like Note::try_new; raises AaError (the reference returns Err) on a bad name."""
        f = ctypes.c_float()
        _lib.check(_lib.load().aa_note_to_freq(name.encode(), 0.0, 440.0, ctypes.byref(f)))
        return cls(name)

    new = try_new

    def get_name(self):
        return self._name

    def get_cents(self):
        return self._cents

    def to_freq(self, base_freq=None):
        """Frequency of the pitch, its cents offset included (theory.rs:168-197)."""
        f = ctypes.c_float()
        _lib.check(_lib.load().aa_note_to_freq(self._name.encode(), float(self._cents),
                                               440.0 if base_freq is None else float(base_freq), ctypes.byref(f)))
        return np.float32(f.value)


class MidiNote:
    """`MidiNote` (theory.rs:6-44)."""

    def __init__(self, midi, cents=0.0):
        self.midi = int(midi)
        self.cents = np.float32(cents)

    @classmethod
    def from_freq(cls, freq, base_freq=None):
        midi = ctypes.c_int32()
        cents = ctypes.c_float()
        _lib.check(_lib.load().aa_midi_from_freq(float(freq), 440.0 if base_freq is None else float(base_freq),
                                                 ctypes.byref(midi), ctypes.byref(cents)))
        return cls(midi.value, cents.value)

    def to_freq(self, base_freq=None):
        f = ctypes.c_float()
        _lib.check(_lib.load().aa_midi_to_freq(self.midi, float(self.cents),
                                               440.0 if base_freq is None else float(base_freq), ctypes.byref(f)))
        return np.float32(f.value)


class Interval:
    """`Interval::new(&freqs, system)` (theory.rs:289-381)."""

    def __init__(self, freqs, system=None):
        f = np.ascontiguousarray(freqs, dtype=np.float32)
        name = ctypes.create_string_buffer(8)
        acc = ctypes.c_float()
        _lib.check(_lib.load().aa_interval(_lib.f32_ptr(f) if len(f) else None, len(f), _system_code(system), name,
                                           ctypes.byref(acc)))
        self._name = name.value.decode()
        self._accuracy = np.float32(acc.value)

    def get_name(self):
        return self._name

    def get_accuracy(self):
        return self._accuracy


class TunerOutput:
    """`struct TunerOutput` (tuner.rs:38-56) as filled by one pass of the tuner loop."""

    def __init__(self, raw, key):
        self._raw = raw
        self.label = raw.label.decode()
        self.cents = np.float32(raw.cents)
        self.notes = [bytes(raw.notes[i]).split(b"\0")[0].decode() for i in range(raw.n_notes)]
        self.accuracies = np.array(raw.accuracies[:raw.n_notes], dtype=np.float32)
        self.mode = [k for k, v in MODES.items() if v == raw.mode][0]
        self.system = [k for k, v in SYSTEMS.items() if v == raw.system][0]
        self.base_freq = np.float32(raw.base_freq)
        self.key = key
        self.beat_position = raw.beat_position

    def to_json(self):
        """The string `Tuner::poll_output` returns (src/lib.rs:80-82)."""
        lib = _lib.load()
        buf = ctypes.create_string_buffer(2048)
        n = ctypes.c_size_t()
        _lib.check(lib.aa_tuner_output_json(ctypes.byref(self._raw), self.key.encode(), buf, len(buf), ctypes.byref(n)))
        return buf.raw[:n.value].decode()


class Tuner:
    """The tuner thread's state and loop body (tuner.rs:60-207) for offline use: `update(freqs,
    scores, beat_position)` is one iteration; an empty list leaves `output` as it was."""

    def __init__(self):
        self.key = "C major"
        self.base = np.float32(440.0)
        self.mode = "MultiPitch"
        self.system = "EqualTemperament"
        # TunerOutput::default(): empty strings and zeros (base_freq 0.0, key "") until the first update
        self.output = TunerOutput(_lib.TunerOutputStruct(), "")

    def set_base_freq(self, freq):
        self.base = np.float32(min(max(np.float32(freq), np.float32(220.0)), np.float32(880.0)))

    def set_key(self, key):
        self.key = key

    def set_mode(self, mode):
        # lib.rs:97-100: anything but "SinglePitch" selects MultiPitch
        self.mode = "SinglePitch" if mode == "SinglePitch" else "MultiPitch"

    def set_system(self, system):
        # lib.rs:107-110: anything but "JustIntonation" selects EqualTemperament
        self.system = "JustIntonation" if system == "JustIntonation" else "EqualTemperament"

    def update(self, freqs, scores, beat_position=0.0):
        f = np.ascontiguousarray(freqs, dtype=np.float32)
        s = np.ascontiguousarray(scores, dtype=np.float32)
        raw = _lib.TunerOutputStruct()
        _lib.check(_lib.load().aa_tuner_output_build(
            _lib.f32_ptr(f) if len(f) else None, _lib.f32_ptr(s) if len(s) else None, len(f), MODES[self.mode],
            SYSTEMS[self.system], float(self.base), float(beat_position), ctypes.byref(raw)))
        if raw.n_notes:
            self.output = TunerOutput(raw, self.key)
        return self.output

    def poll_output(self):
        return self.output.to_json()

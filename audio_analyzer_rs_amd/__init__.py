"""MI355X-native STFT / spectrogram hot path — drop-in for audio-analyzer-rs's
`FftProcessor` + frame loop (src/dsp/fft.rs, src/audio_io/stft.rs:273-318,
src/analysis/onset.rs:244-272).

Everything computes through libaa_stft.so (HIP kernels for gfx950 behind the C ABI of
include/aa_stft.h).  Importing the package does not need a GPU; creating a plan does.
"""
from . import _lib
from ._lib import AaError, LIB_PATH  # noqa: F401
from .fft import FftProcessor, hann_window  # noqa: F401
from .stft import (Conditioner, PitchTracker, Stft, StftStream, downmix, global_floor_per_frame, onset_flux, onset_gates, pin, tuner_floor,  # noqa: F401
                   tuner_pitches, unpin)
from .sharding import (Comm, IpcTransport, ShardPlan, gather_rows, gather_samples, overlap_counts, overlapped_e2e, piece_samples, scatter_samples,  # noqa: F401
                       set_transport, shard_frames, shard_piece, shard_samples)
from . import signals, theory, wav  # noqa: F401
from .theory import Interval, MidiNote, Note, Tuner, TunerOutput  # noqa: F401


def load_library():
    """Load (and return) the ctypes handle of libaa_stft.so; raises if it is not built."""
    return _lib.load()

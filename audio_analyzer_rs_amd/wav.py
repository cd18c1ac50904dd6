"""The on-disk format either side of the path ("next" row f-4, input half): the WAV files the
reference's recorder writes (src/audio_io/recorder.rs:60-85: 16-bit PCM, mono, the stream's sample
rate, each sample `(clamp(s, -1, 1) * 32767) as i16`) and how a decoder hands them back
(`i16 / 32768`, what symphonia's i16 -> f32 conversion does in src/audio_io/player.rs).

Plain RIFF/WAVE parsing with the standard library; the samples go to the GPU as float32.
"""
import struct
import wave

import numpy as np


def encode_pcm16(samples):
    """float32 [-1, 1] -> int16 exactly as recorder.rs:81-83: clamp, scale by i16::MAX, `as i16`
    (truncation toward zero; `s.max(-1.0)` turns a NaN into -1.0)."""
    s = np.asarray(samples, dtype=np.float32)
    s = np.where(np.isnan(s), np.float32(-1.0), s)
    return np.trunc(np.clip(s, np.float32(-1.0), np.float32(1.0)) * np.float32(32767.0)).astype(np.int16)


def decode_pcm16(q):
    """int16 -> float32 in [-1, 1): i16 / 32768."""
    return (np.asarray(q, dtype=np.int16).astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def write_wav(path, samples, sample_rate=48000):
    """Write mono 16-bit PCM like `rec start` does."""
    q = encode_pcm16(samples)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(q.astype("<i2").tobytes())


def read_wav(path):
    """Read a 16-bit PCM WAV file.  Returns (float32 samples, sample_rate): mono files as a
    vector; multi-channel files as [frames, channels] interleaved-order rows (downmix with
    audio_analyzer_rs_amd.downmix, the capture rule of audio_io/mod.rs:784-794)."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2 or w.getcomptype() != "NONE":
            raise ValueError(f"{path}: only 16-bit PCM is supported (the recorder's format)")
        ch, sr, n = w.getnchannels(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    q = np.frombuffer(raw, dtype="<i2")
    x = decode_pcm16(q)
    return (x if ch == 1 else x.reshape(-1, ch)), sr


def onsets_json(beat_positions, raw_sample_offsets, velocities):
    """The string `OnsetDetection::poll_onsets` returns (src/lib.rs:326-338): a JSON array of
    {"beat_position": %.6f, "raw_sample_offset": %d, "velocity": %.4f} objects."""
    items = [
        '{"beat_position":%.6f,"raw_sample_offset":%d,"velocity":%.4f}' % (float(b), int(o), float(v))
        for b, o, v in zip(beat_positions, raw_sample_offsets, velocities)
    ]
    return "[" + ",".join(items) + "]"

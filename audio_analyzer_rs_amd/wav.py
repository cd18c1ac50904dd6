"""The on-disk format either side of the path ("next" row f-4, input half): the WAV files the
reference's recorder writes (src/audio_io/recorder.rs:60-85: 16-bit PCM, mono, the stream's sample
rate, each sample `(clamp(s, -1, 1) * 32767) as i16`) and how a decoder hands them back
(`i16 / 32768`, what symphonia's i16 -> f32 conversion does in src/audio_io/player.rs).

Plain RIFF/WAVE parsing; the samples go to the GPU as float32.  read_wav also takes the other linear WAV sample
formats a general decoder (the reference's player) accepts.
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


# sample formats of RIFF/WAVE `fmt ` chunks: 1 = integer PCM, 3 = IEEE float, 0xFFFE = WAVE_FORMAT_EXTENSIBLE (the real
# format is the first two bytes of the sub-format GUID)
_PCM, _FLOAT, _EXTENSIBLE = 1, 3, 0xFFFE


def decode_samples(raw, fmt, bits):
    """Interleaved little-endian sample bytes -> float32 in [-1, 1), with the scale factors a general decoder applies
    (the reference plays files back through symphonia, src/generators/player.rs:171-232, whose f32 sample buffer
    holds u8 as (x - 128) / 128, i16 / 2^15, i24 / 2^23, i32 / 2^31 and floats as they are; symphonia itself is not
    in the tree, so these are its published conversions: parity unpinned).  16-bit PCM is the recorder's format."""
    if fmt == _PCM and bits == 8:
        return ((np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)).astype(np.float32)
    if fmt == _PCM and bits == 16:
        return decode_pcm16(np.frombuffer(raw, dtype="<i2"))
    if fmt == _PCM and bits == 24:
        # the three bytes into the upper three of a little-endian int32, an arithmetic shift for the sign; in pieces, so
        # that an hour of audio (518 MB of samples) needs the output and one piece of temporaries, not ~30 B per sample
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3)
        out = np.empty(b.shape[0], dtype=np.float32)
        step = 1 << 22
        for lo in range(0, b.shape[0], step):
            piece = b[lo:lo + step]
            q = np.zeros((piece.shape[0], 4), dtype=np.uint8)
            q[:, 1:] = piece
            out[lo:lo + step] = ((q.view("<i4")[:, 0] >> 8).astype(np.float64) / 8388608.0).astype(np.float32)
        return out
    if fmt == _PCM and bits == 32:
        return (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    if fmt == _FLOAT and bits == 32:
        return np.frombuffer(raw, dtype="<f4").astype(np.float32)
    if fmt == _FLOAT and bits == 64:
        return np.frombuffer(raw, dtype="<f8").astype(np.float32)
    raise ValueError(f"unsupported WAV sample format (format tag {fmt}, {bits} bits)")


def read_wav(path):
    """Read a RIFF/WAVE file: integer PCM of 8 / 16 / 24 / 32 bits or IEEE float of 32 / 64 bits, plain or
    WAVE_FORMAT_EXTENSIBLE header, any channel count.  Returns (float32 samples, sample_rate): mono files as a
    vector; multi-channel files as [frames, channels] interleaved-order rows (downmix with
    audio_analyzer_rs_amd.downmix, the capture rule of audio_io/mod.rs:784-794).  16-bit PCM mono is what the
    reference's recorder writes; the other formats are what its player accepts (decode_samples)."""
    with open(str(path), "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    fmt = ch = sr = bits = align = None
    pos = 12
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            if len(body) < 16:
                raise ValueError(f"{path}: short fmt chunk")
            fmt, ch, sr, _, align, bits = struct.unpack_from("<HHIIHH", body, 0)
            if fmt == _EXTENSIBLE:
                if len(body) < 26:
                    raise ValueError(f"{path}: short WAVE_FORMAT_EXTENSIBLE chunk")
                fmt = struct.unpack_from("<H", body, 24)[0]            # first two bytes of the sub-format GUID
        elif tag == b"data":
            if fmt is None:
                raise ValueError(f"{path}: data chunk in front of the fmt chunk")
            if ch < 1 or bits < 8 or bits % 8 or align != ch * bits // 8:
                raise ValueError(f"{path}: inconsistent fmt chunk ({ch} channels, {bits} bits, block align {align})")
            whole = len(body) - len(body) % align                      # (a truncated last frame is dropped)
            x = decode_samples(body[:whole], fmt, bits)
            return (x if ch == 1 else x.reshape(-1, ch)), sr
        pos += 8 + size + (size & 1)
    raise ValueError(f"{path}: no data chunk")


def onsets_json(beat_positions, raw_sample_offsets, velocities):
    """The string `OnsetDetection::poll_onsets` returns (src/lib.rs:326-338): a JSON array of
    {"beat_position": %.6f, "raw_sample_offset": %d, "velocity": %.4f} objects."""
    items = [
        '{"beat_position":%.6f,"raw_sample_offset":%d,"velocity":%.4f}' % (float(b), int(o), float(v))
        for b, o, v in zip(beat_positions, raw_sample_offsets, velocities)
    ]
    return "[" + ",".join(items) + "]"

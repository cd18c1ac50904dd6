"""Batched STFT: the frame loop of `STFT::detect_pitches` (src/audio_io/stft.rs:273-318)
and `OnsetDetector::detect_onsets` (src/analysis/onset.rs:244-272) over a whole buffer.

    plan = Stft(n_fft=2048, hop=512)             # the tuner geometry (stft.rs:169-170)
    mags = plan(samples)                         # [n_frames, n_fft/2+1], |X| = c.norm()

`samples` may be a numpy float32 array (host path: staged through the plan's device
workspace) or a CUDA torch.float32 tensor (device path: zero-copy, asynchronous on the
current torch stream).  Frame t covers samples[t*hop : t*hop + n_fft]; no centring, no
padding, the trailing partial frame is dropped — exactly the reference's loop.
"""
import ctypes

import numpy as np

from . import _lib


def _current_device_index():
    """The HIP device a plan created now binds to (aa_stft_plan_create uses the current one)."""
    try:
        import torch

        return torch.cuda.current_device() if torch.cuda.is_available() else 0
    except ImportError:
        return 0


class _RowsHolder:
    """Owner of an aa_rows_alloc buffer, seen by torch through __cuda_array_interface__ (torch keeps the holder alive for
    as long as a tensor views the memory; the buffer goes back with the holder)."""

    def __init__(self, lib, ptr, shape):
        self._lib, self._ptr = lib, ptr
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2,
                                         "strides": None}

    def __del__(self):
        try:
            if self._ptr:
                self._lib.aa_rows_free(self._ptr)
                self._ptr = None
        except Exception:
            pass


class Stft:
    def __init__(self, n_fft, hop=None, window="hann", output="mag", sample_rate=48000,
                 n_mels=128, n_mfcc=40, fmin=0.0, fmax=0.0, log_floor=0.0):
        self._lib = _lib.load()
        hop = n_fft // 4 if hop is None else hop       # hop = N/4 in both reference instances
        custom = None
        if isinstance(window, str):
            wkind = _lib.WINDOW_KINDS[window]
        else:
            custom = np.ascontiguousarray(window, dtype=np.float32)
            if custom.shape != (n_fft,):
                raise _lib.AaError(_lib.AA_ERR_LEN, "custom window must have n_fft entries")
            wkind = _lib.AA_WINDOW_CUSTOM
        cfg = _lib.StftConfig(int(n_fft), int(hop), wkind, _lib.OUTPUT_KINDS[output],
                              int(sample_rate), int(n_mels), int(n_mfcc), float(fmin), float(fmax),
                              float(log_floor))
        self._plan = ctypes.c_void_p()
        _lib.check(self._lib.aa_stft_plan_create(
            ctypes.byref(cfg), _lib.f32_ptr(custom) if custom is not None else None,
            ctypes.byref(self._plan)))
        self.n_fft, self.hop, self.output = int(n_fft), int(hop), output
        self.cols = int(self._lib.aa_stft_out_cols(self._plan))
        self.kernel = self._lib.aa_stft_kernel_name(self._plan).decode()
        self.mel_path = self._lib.aa_stft_mel_path_name(self._plan).decode()
        self.device_index = _current_device_index()

    def close(self):
        if getattr(self, "_plan", None):
            self._lib.aa_stft_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_frames(self, n_samples):
        """floor((L - N)/hop) + 1 (stft.rs:273, 436-437); integer-exact."""
        return int(self._lib.aa_stft_num_frames(self._plan, int(n_samples)))

    def alloc_rows(self, n_frames, candidates=128):
        """A CUDA float32 row buffer [n_frames, cols] placed by measurement (aa_rows_alloc: slab after slab - up to
        `candidates` - with the plan's launch timed into each, until one is clearly faster than the others) - for the
        complex rows, whose launch time depends on the physical memory behind the buffer.  candidates = 1: a plain
        allocation.  Returns (tensor, kept_us, typical_us); the memory is released with the tensor."""
        import torch

        ptr = ctypes.c_void_p()
        kept, typical = ctypes.c_float(0.0), ctypes.c_float(0.0)
        _lib.check(self._lib.aa_rows_alloc(self._plan, int(n_frames), int(candidates), ctypes.byref(ptr),
                                           ctypes.byref(kept), ctypes.byref(typical)))
        holder = _RowsHolder(self._lib, ptr.value, (int(n_frames), self.cols))
        t = torch.as_tensor(holder, device=torch.device("cuda", self.device_index))
        return t, float(kept.value), float(typical.value)

    def _wrap(self, flat, nf):
        if self.output == "complex":
            return flat.reshape(nf, self.cols // 2, 2)
        return flat.reshape(nf, self.cols)

    def __call__(self, samples, frame_begin=0, frame_end=None, out=None, stream=None):
        if isinstance(samples, np.ndarray):
            if frame_begin != 0 or frame_end is not None or stream is not None:
                raise _lib.AaError(_lib.AA_ERR_ARG, "the host path transforms the whole buffer: "
                                   "frame_begin / frame_end / stream apply to CUDA tensors only")
            return self.execute_host(samples, out)
        return self.execute_device(samples, frame_begin, frame_end, out, stream)

    def execute_host(self, samples, out=None):
        x = np.ascontiguousarray(samples, dtype=np.float32)
        if x.ndim != 1:
            raise _lib.AaError(_lib.AA_ERR_ARG, "samples must be one-dimensional (mono)")
        nf = self.num_frames(x.shape[0])
        if out is None:
            out = np.empty((nf, self.cols), dtype=np.float32)
        elif out.dtype != np.float32 or out.shape != (nf, self.cols) or not out.flags.c_contiguous:
            raise _lib.AaError(_lib.AA_ERR_ARG, f"out must be C-contiguous float32 [{nf}, {self.cols}]")
        if nf:
            _lib.check(self._lib.aa_stft_execute_host(self._plan, _lib.f32_ptr(x), x.shape[0],
                                                      _lib.f32_ptr(out)))
        return out.view(np.complex64) if self.output == "complex" else out

    def execute_device(self, samples, frame_begin=0, frame_end=None, out=None, stream=None):
        """samples: CUDA float32 1-D tensor. Returns a CUDA tensor [frames, cols]
        (complex64 [frames, n_fft/2+1] for output="complex")."""
        import torch

        if not (samples.is_cuda and samples.dtype == torch.float32 and samples.dim() == 1
                and samples.is_contiguous()):
            raise _lib.AaError(_lib.AA_ERR_ARG, "samples must be a contiguous 1-D CUDA float32 tensor")
        total = self.num_frames(samples.shape[0])
        if frame_end is None:
            frame_end = total
        if not 0 <= frame_begin <= frame_end <= total:
            raise _lib.AaError(_lib.AA_ERR_ARG, f"frame range [{frame_begin}, {frame_end}) is outside "
                               f"[0, {total}]")
        if samples.device.index != self.device_index:
            raise _lib.AaError(_lib.AA_ERR_ARG, f"samples live on cuda:{samples.device.index} but the "
                               f"plan was created on cuda:{self.device_index}")
        nf = frame_end - frame_begin
        if out is None:
            out = torch.empty((max(nf, 0), self.cols), dtype=torch.float32, device=samples.device)
        elif not (out.is_cuda and out.device == samples.device and out.dtype == torch.float32
                  and out.is_contiguous() and out.numel() >= nf * self.cols):
            # the kernels write nf*cols floats with 16-byte stores: a short or strided `out`
            # would be a silent out-of-bounds device write
            raise _lib.AaError(_lib.AA_ERR_ARG, f"out must be a contiguous CUDA float32 tensor on "
                               f"{samples.device} with at least {nf * self.cols} elements")
        st = stream if stream is not None else torch.cuda.current_stream(samples.device).cuda_stream
        if nf > 0:
            _lib.check(self._lib.aa_stft_execute_device(self._plan, samples.data_ptr(),
                                                        samples.shape[0], frame_begin, frame_end,
                                                        out.data_ptr(), st))
        if self.output == "complex":
            return torch.view_as_complex(out.view(-1, self.cols // 2, 2))
        return out


    def execute_channels_device(self, samples, out=None, stream=None):
        """samples: contiguous CUDA float32 [channels, n_samples] (planar, equal-length mono channels:
        BASELINE configs[2] is two of them).  Returns [channels, frames, cols] (complex64
        [channels, frames, n_fft/2+1] for output="complex"); every channel's rows are bit-identical to
        `execute_device` on that channel.  One launch for the wave kernels of n_fft 1024 / 2048 when
        n_samples and frames * cols are multiples of 4, else channel after channel
        (aa_stft_execute_channels_device)."""
        import torch

        if not (samples.is_cuda and samples.dtype == torch.float32 and samples.dim() == 2
                and samples.is_contiguous()):
            raise _lib.AaError(_lib.AA_ERR_ARG, "samples must be a contiguous [channels, n] CUDA float32 tensor")
        if samples.device.index != self.device_index:
            raise _lib.AaError(_lib.AA_ERR_ARG, f"samples live on cuda:{samples.device.index} but the "
                               f"plan was created on cuda:{self.device_index}")
        ch, n = int(samples.shape[0]), int(samples.shape[1])
        nf = self.num_frames(n)
        if out is None:
            out = torch.empty((ch, nf, self.cols), dtype=torch.float32, device=samples.device)
        elif not (out.is_cuda and out.device == samples.device and out.dtype == torch.float32
                  and out.is_contiguous() and tuple(out.shape) == (ch, nf, self.cols)):
            raise _lib.AaError(_lib.AA_ERR_ARG, f"out must be a contiguous CUDA float32 tensor of shape "
                               f"{(ch, nf, self.cols)} on {samples.device}")
        st = stream if stream is not None else torch.cuda.current_stream(samples.device).cuda_stream
        if ch and nf:
            _lib.check(self._lib.aa_stft_execute_channels_device(self._plan, samples.data_ptr(), n, n, ch,
                                                                 out.data_ptr(), nf * self.cols, st))
        if self.output == "complex":
            return torch.view_as_complex(out.view(ch, -1, self.cols // 2, 2))
        return out


def pin(array):
    """Page-lock a numpy array in place (aa_host_register).  With both the samples and the output
    array pinned, `Stft.execute_host` overlaps H2D, kernels and D2H on two streams."""
    lib = _lib.load()
    _lib.check(lib.aa_host_register(array.ctypes.data, array.nbytes))
    return array


def unpin(array):
    lib = _lib.load()
    _lib.check(lib.aa_host_unregister(array.ctypes.data))


def downmix(interleaved, channels, stream=None):
    """Average the first min(channels, 2) channels of an interleaved CUDA float32 tensor to
    mono — what the capture callback does before the STFT workers see a slot
    (src/audio_io/mod.rs:784-794)."""
    import torch

    lib = _lib.load()
    if not (interleaved.is_cuda and interleaved.dtype == torch.float32 and interleaved.is_contiguous()):
        raise _lib.AaError(_lib.AA_ERR_ARG, "interleaved must be a contiguous CUDA float32 tensor")
    n = interleaved.numel() // int(channels)
    out = torch.empty(n, dtype=torch.float32, device=interleaved.device)
    st = stream if stream is not None else torch.cuda.current_stream(interleaved.device).cuda_stream
    if n:
        _lib.check(lib.aa_downmix_device(interleaved.data_ptr(), n, int(channels), out.data_ptr(), st))
    return out


_OFFSETS_SEEN = {}          # the batch description checked last: a weak reference to the tensor, its version, the frame count


def _offsets_arg(offsets, nf):
    """Validate a batch description: CUDA int64 [n_streams + 1], 0 .. total frames.  The check reads the tensor back (a
    device synchronisation): it is done once per tensor version, so that a chain of calls over the same streams stays
    asynchronous."""
    import torch

    if not (offsets.is_cuda and offsets.dtype == torch.int64 and offsets.dim() == 1 and offsets.is_contiguous()
            and offsets.numel() >= 2):
        raise _lib.AaError(_lib.AA_ERR_ARG, "offsets must be a contiguous CUDA int64 vector [n_streams + 1]")
    seen = _OFFSETS_SEEN.get("last")
    if not (seen and seen[0]() is offsets and seen[1:] == (offsets._version, int(nf))):
        host = offsets.cpu()
        if int(host[0]) != 0 or int(host[-1]) != nf or bool((host[1:] < host[:-1]).any()):
            raise _lib.AaError(_lib.AA_ERR_ARG, "offsets must ascend from 0 to the number of frames")
        import weakref

        _OFFSETS_SEEN["last"] = (weakref.ref(offsets), offsets._version, int(nf))   # (the very tensor object, unchanged)
    return offsets.numel() - 1


def onset_flux(mags, prev_row=None, stream=None, offsets=None):
    """Spectral flux and frame energy of the onset detector's front end
    (src/analysis/onset.rs:261-291) for a CUDA float32 magnitude spectrogram [frames, bins].
    offsets (CUDA int64 [n_streams + 1]): the frames are that many independent streams laid end to
    end; prev_row is then [n_streams, bins] (or None).
    Returns (flux, energy), two CUDA float32 vectors of length frames."""
    import torch

    lib = _lib.load()
    if not (mags.is_cuda and mags.dtype == torch.float32 and mags.dim() == 2 and mags.is_contiguous()):
        raise _lib.AaError(_lib.AA_ERR_ARG, "mags must be a contiguous 2-D CUDA float32 tensor")
    nf, nb = mags.shape
    flux = torch.empty(nf, dtype=torch.float32, device=mags.device)
    energy = torch.empty(nf, dtype=torch.float32, device=mags.device)
    st = stream if stream is not None else torch.cuda.current_stream(mags.device).cuda_stream
    if nf and offsets is not None:
        ns = _offsets_arg(offsets, nf)
        _lib.check(lib.aa_onset_flux_batch_device(mags.data_ptr(), offsets.data_ptr(), ns, nf, nb,
                                                  prev_row.data_ptr() if prev_row is not None else None,
                                                  flux.data_ptr(), energy.data_ptr(), st))
    elif nf:
        _lib.check(lib.aa_onset_flux_device(mags.data_ptr(), nf, nb,
                                            prev_row.data_ptr() if prev_row is not None else None,
                                            flux.data_ptr(), energy.data_ptr(), st))
    return flux, energy


ONSET_FLAG_FLUX, ONSET_FLAG_BURST, ONSET_FLAG_DETECTED, ONSET_FLAG_ENERGY_RISING, ONSET_FLAG_FIRED = 1, 2, 4, 8, 16
ONSET_FLAG_TICK, ONSET_FLAG_GATED, ONSET_FLAG_CAL_CONSUMED = 32, 64, 128
ONSET_MASK_TICK, ONSET_MASK_CAL_OPEN, ONSET_MASK_CAL_TIMEOUT = 1, 2, 4
ONSET_UNCALIBRATED, ONSET_CALIBRATED, ONSET_CALIBRATING = 0, 1, 2


def onset_gates(mags, flux, energy, global_floor, state=None, stream=None, offsets=None, suppress=None,
                calibrated=True, initialized=None):
    """The onset detector's per-frame logic after the flux (src/analysis/onset.rs:298-371,
    531-535): burst floor, silence gate, energy EMA, FluxTracker, combined trigger, velocity,
    frames_since_onset gate.  All arguments are CUDA tensors: mags float32 [frames, bins]; flux,
    energy, global_floor float32 [frames].  The transport's inputs: suppress (uint8 [frames]: bit 0 = inside
    the metronome's tick guard, onset.rs:393-396; bit 1 = calibration click played and the residual of an onset
    at this frame plausible, :405-422; bit 2 = calibration timed out, :361-371; None = 0) and calibrated
    (False / 0 = calibration running, click not played: nothing fires; True / 1 = done; 2 = calibrating:
    completion is carried in the state, the first gated onset on a bit-1 frame completes it and is consumed,
    flags bit 7, onset.rs:404-440).
    state: None for a fresh detector, or the [bins + 4] tensor returned by a previous call.
    offsets (CUDA int64 [n_streams + 1]): independent streams laid end to end; state is then
    [n_streams, bins + 4] and initialized (CUDA int32 [n_streams], default: all ones when a state is
    passed) says which streams carry one - a stream that has had no frame yet stays fresh.
    Returns (dict of per-frame tensors, state)."""
    import torch

    lib = _lib.load()
    nf, nb = mags.shape
    have_state = state is not None
    ns = _offsets_arg(offsets, nf) if offsets is not None else None
    if state is None:
        state = torch.zeros(nb + 4 if ns is None else (ns, nb + 4), dtype=torch.float32, device=mags.device)
    elif tuple(state.shape) != ((nb + 4,) if ns is None else (ns, nb + 4)):
        raise ValueError("state must hold bins + 4 floats per stream")
    out = {
        "flux": torch.empty(nf, dtype=torch.float32, device=mags.device),
        "energy_ema": torch.empty(nf, dtype=torch.float32, device=mags.device),
        "max_excess": torch.empty(nf, dtype=torch.float32, device=mags.device),
        "burst_count": torch.empty(nf, dtype=torch.int32, device=mags.device),
        "flags": torch.empty(nf, dtype=torch.uint8, device=mags.device),
        "velocity": torch.empty(nf, dtype=torch.float32, device=mags.device),
    }
    st = stream if stream is not None else torch.cuda.current_stream(mags.device).cuda_stream
    if suppress is not None and (suppress.dtype != torch.uint8 or suppress.numel() != nf or not suppress.is_cuda):
        raise ValueError("suppress must be a CUDA uint8 tensor with one entry per frame")
    sup = suppress.data_ptr() if suppress is not None else None
    cal = int(calibrated)
    if cal not in (0, 1, 2):
        raise ValueError("calibrated must be 0 / False, 1 / True or 2")
    if nf and ns is not None:
        if initialized is None:
            init = torch.full((ns,), 1 if have_state else 0, dtype=torch.int32, device=mags.device)
        else:
            init = initialized.to(device=mags.device, dtype=torch.int32).contiguous()
            if init.numel() != ns:
                raise ValueError("initialized must hold one int32 per stream")
        _lib.check(lib.aa_onset_gates_batch_device(
            mags.data_ptr(), offsets.data_ptr(), ns, nf, nb, flux.data_ptr(), energy.data_ptr(),
            global_floor.data_ptr(), sup, cal, state.data_ptr(), init.data_ptr(), out["flux"].data_ptr(),
            out["energy_ema"].data_ptr(), out["max_excess"].data_ptr(), out["burst_count"].data_ptr(),
            out["flags"].data_ptr(), out["velocity"].data_ptr(), st))
    elif nf:
        _lib.check(lib.aa_onset_gates_device(
            mags.data_ptr(), nf, nb, flux.data_ptr(), energy.data_ptr(), global_floor.data_ptr(), sup,
            cal, state.data_ptr(), 1 if have_state else 0, out["flux"].data_ptr(),
            out["energy_ema"].data_ptr(),
            out["max_excess"].data_ptr(), out["burst_count"].data_ptr(), out["flags"].data_ptr(),
            out["velocity"].data_ptr(), st))
    elif not have_state:
        state = None                                  # nothing seen: the detector is still fresh
    return out, state


def tuner_floor(mags, global_floor, state=None, stream=None, offsets=None, initialized=None):
    """Per-bin adaptive noise floor of the tuner (src/audio_io/stft.rs:326-367) for a CUDA float32
    magnitude spectrogram [frames, bins].  global_floor: CUDA float32 [frames]
    (= 10^(noise_floor_db/20) * bins / 2 per frame).  state: None for a fresh stream, or the
    [3, bins] tensor returned by a previous call.  offsets (CUDA int64 [n_streams + 1]): independent streams laid
    end to end; state is then [n_streams, 3, bins] and initialized (CUDA int32 [n_streams], default: all ones when
    a state is passed) says which streams carry one - a stream that has had no frame yet stays fresh.
    Returns (effective_floor, state)."""
    import torch

    lib = _lib.load()
    nf, nb = mags.shape
    per_stream, initialized = initialized, state is not None
    ns = _offsets_arg(offsets, nf) if offsets is not None else None
    if state is None:
        state = torch.zeros((3, nb) if ns is None else (ns, 3, nb), dtype=torch.float32, device=mags.device)
    eff = torch.empty_like(mags)
    st = stream if stream is not None else torch.cuda.current_stream(mags.device).cuda_stream
    if nf and ns is not None:
        if per_stream is None:
            init = torch.full((ns,), 1 if initialized else 0, dtype=torch.int32, device=mags.device)
        else:
            init = per_stream.to(device=mags.device, dtype=torch.int32).contiguous()
            if init.numel() != ns:
                raise ValueError("initialized must hold one int32 per stream")
        _lib.check(lib.aa_tuner_floor_batch_device(mags.data_ptr(), offsets.data_ptr(), ns, nf, nb,
                                                   global_floor.data_ptr(), state.data_ptr(), init.data_ptr(),
                                                   eff.data_ptr(), st))
    elif nf:
        _lib.check(lib.aa_tuner_floor_device(mags.data_ptr(), nf, nb, global_floor.data_ptr(),
                                             state.data_ptr(), 1 if initialized else 0, eff.data_ptr(), st))
    return eff, state


def tuner_pitches(mags, eff_floor, bin_width, min_freq=24.0, max_freq=10000.0, stream=None):
    """Multi-pitch extraction of the tuner (src/audio_io/stft.rs:443-620) for every frame of a CUDA
    float32 magnitude spectrogram [frames, bins] and its effective noise floor (tuner_floor).
    Returns (freq [frames, 8], score [frames, 8], count [frames]) CUDA tensors."""
    import torch

    lib = _lib.load()
    nf, nb = mags.shape
    freq = torch.empty((nf, 8), dtype=torch.float32, device=mags.device)
    score = torch.empty((nf, 8), dtype=torch.float32, device=mags.device)
    count = torch.empty(nf, dtype=torch.int32, device=mags.device)
    st = stream if stream is not None else torch.cuda.current_stream(mags.device).cuda_stream
    if nf:
        _lib.check(lib.aa_tuner_pitches_device(mags.data_ptr(), eff_floor.data_ptr(), nf, nb, float(bin_width),
                                               float(min_freq), float(max_freq), freq.data_ptr(),
                                               score.data_ptr(), count.data_ptr(), st))
    return freq, score, count


class Conditioner:
    """Mirror of the reducer thread's conditioning (src/audio_io/mod.rs:351-478 + DynamicsTracker,
    dynamics.rs): biquads, gate, AGC per slot, on the host.  process() takes whole slots (numpy
    float32), returns the conditioned samples and the per-slot DynamicsOutput arrays."""

    def __init__(self, sample_rate=48000.0, slot_len=1024):
        self._lib = _lib.load()
        self.slot_len = int(slot_len)
        h = ctypes.c_void_p()
        _lib.check(self._lib.aa_conditioner_create(float(sample_rate), self.slot_len, ctypes.byref(h)))
        self._h = h

    def process(self, samples):
        x = np.array(samples, dtype=np.float32)
        n_slots = len(x) // self.slot_len
        x = x[: n_slots * self.slot_len].copy()
        out = {"noise_floor_db": np.empty(n_slots, np.float32), "rms_db": np.empty(n_slots, np.float32),
               "gain_db": np.empty(n_slots, np.float32), "session_median_db": np.empty(n_slots, np.float32),
               "level": np.empty(n_slots, np.int32)}
        _lib.check(self._lib.aa_conditioner_process(
            self._h, x.ctypes.data, n_slots, out["noise_floor_db"].ctypes.data, out["rms_db"].ctypes.data,
            out["gain_db"].ctypes.data, out["session_median_db"].ctypes.data, out["level"].ctypes.data))
        return x, out

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.aa_conditioner_destroy(self._h)
            self._h = None


def global_floor_per_frame(noise_floor_db, n_frames, n_fft, hop, n_bins, slot_len=1024):
    """global_floor[t] = 10^(noise_floor_db / 20) * n_bins / 2 (stft.rs:322-324, onset.rs:299-300)
    with the value of the slot that holds frame t's last sample - the latest one the reference's
    worker can have read when it processes that frame.  float32 numpy [n_frames]."""
    nf_db = np.asarray(noise_floor_db, dtype=np.float32)
    last = (np.arange(n_frames, dtype=np.int64) * hop + n_fft - 1) // slot_len
    last = np.minimum(last, len(nf_db) - 1)
    return (np.float32(10.0) ** (nf_db[last] / np.float32(20.0)) * np.float32(n_bins) / np.float32(2.0)).astype(np.float32)


class PitchTracker:
    """Mirror of the reference's PitchTracker (src/audio_io/stft.rs:28-117): feed it the per-frame
    raw pitches of tuner_pitches (numpy arrays on the host), get the stable pitches back."""

    TRACKS_MAX = 32

    def __init__(self):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._lib.aa_pitch_tracker_create(ctypes.byref(h)))
        self._h = h

    def process(self, freq, score, count, onset=None):
        freq = np.ascontiguousarray(freq, dtype=np.float32)
        score = np.ascontiguousarray(score, dtype=np.float32)
        count = np.ascontiguousarray(count, dtype=np.int32)
        nf = freq.shape[0]
        on = None if onset is None else np.ascontiguousarray(onset, dtype=np.uint8)
        of = np.zeros((nf, self.TRACKS_MAX), np.float32)
        os_ = np.zeros((nf, self.TRACKS_MAX), np.float32)
        oc = np.zeros(nf, np.int32)
        _lib.check(self._lib.aa_pitch_tracker_process(
            self._h, freq.ctypes.data, score.ctypes.data, count.ctypes.data, None if on is None else on.ctypes.data,
            nf, of.ctypes.data, os_.ctypes.data, oc.ctypes.data))
        return of, os_, oc

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.aa_pitch_tracker_destroy(self._h)
            self._h = None


class StftStream:
    """Streaming front end: the ring-buffer ingest of stft.rs:198-201, 240-266 feeding the
    GPU frame loop.  `push(slot)` appends one slot (the reference's are 1024 samples,
    src/audio_io/mod.rs:126-128) and returns (rows, frame_starts) for every completed frame."""

    def __init__(self, n_fft, hop=None, ring_len=0, output="mag", window="hann"):
        self._lib = _lib.load()
        hop = n_fft // 4 if hop is None else hop
        cfg = _lib.StftConfig(int(n_fft), int(hop), _lib.WINDOW_KINDS[window],
                              _lib.OUTPUT_KINDS[output], 48000, 0, 0, 0.0, 0.0, 0.0)
        self._s = ctypes.c_void_p()
        _lib.check(self._lib.aa_stream_create(ctypes.byref(cfg), int(ring_len), ctypes.byref(self._s)))
        self.n_fft, self.hop, self.output = int(n_fft), int(hop), output
        self.ring_len = int(ring_len) if ring_len else max(8192, 4 * int(n_fft))
        nb = n_fft // 2 + 1
        self.cols = 2 * nb if output == "complex" else nb

    def close(self):
        if getattr(self, "_s", None):
            self._lib.aa_stream_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, slot):
        slot = np.ascontiguousarray(slot, dtype=np.float32)
        cap = slot.shape[0] // self.hop + self.ring_len // self.hop + 2
        out = np.empty((cap, self.cols), dtype=np.float32)
        starts = np.empty(cap, dtype=np.int64)
        n = ctypes.c_int64(0)
        _lib.check(self._lib.aa_stream_push(
            self._s, _lib.f32_ptr(slot), slot.shape[0], _lib.f32_ptr(out),
            starts.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), cap, ctypes.byref(n)))
        rows = out[:n.value]
        if self.output == "complex":
            rows = rows.view(np.complex64)
        return rows, starts[:n.value]

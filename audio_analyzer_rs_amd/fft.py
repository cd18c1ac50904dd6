"""Host-side mirror of the reference's `FftProcessor` (src/dsp/fft.rs:6-42).

Same names and argument meaning as the Rust type so parity tests read like the
reference's call sites (src/audio_io/stft.rs:180, 306; src/analysis/onset.rs:141, 259):

    fft = FftProcessor(2048)
    spectrum = fft.process_forward(windowed)     # complex64[len/2 + 1], unnormalised
    signal   = fft.process_inverse(spectrum)     # float32[len], unnormalised (len * x)

Every call goes through the C ABI (include/aa_stft.h) to the HIP kernels.
"""
import ctypes

import numpy as np

from . import _lib


def hann_window(n):
    """`STFT::hann_window(n)` (src/audio_io/stft.rs:641-648): periodic Hann, all f32."""
    lib = _lib.load()
    w = np.empty(int(n), dtype=np.float32)
    _lib.check(lib.aa_hann_window(int(n), _lib.f32_ptr(w)))
    return w


class FftProcessor:
    """Forward / inverse real FFT of one frame, plan-owned device buffers."""

    def __init__(self, length):
        self._lib = _lib.load()
        self._plan = ctypes.c_void_p()
        # The reference panics on an unsupported length (fft.rs:69); here: AaError(AA_ERR_LEN).
        _lib.check(self._lib.aa_fft_create(int(length), ctypes.byref(self._plan)))
        self.len = int(length)

    def close(self):
        if getattr(self, "_plan", None):
            self._lib.aa_fft_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_forward(self, windowed):
        """fft.rs:33-35 — R2C of `windowed` (len f32) -> complex64[len/2+1]."""
        x = np.ascontiguousarray(windowed, dtype=np.float32)
        if x.ndim != 1 or x.shape[0] != self.len:
            raise _lib.AaError(_lib.AA_ERR_LEN, f"input length {x.shape} != plan length {self.len}")
        out = np.empty(self.len + 2, dtype=np.float32)
        _lib.check(self._lib.aa_fft_forward(self._plan, _lib.f32_ptr(x), _lib.f32_ptr(out)))
        return out.view(np.complex64)

    def process_inverse(self, spectrum):
        """fft.rs:39-41 — C2R of complex64[len/2+1] -> float32[len] (unnormalised)."""
        s = np.ascontiguousarray(spectrum, dtype=np.complex64)
        if s.ndim != 1 or s.shape[0] != self.len // 2 + 1:
            raise _lib.AaError(_lib.AA_ERR_LEN,
                               f"spectrum length {s.shape} != {self.len // 2 + 1}")
        out = np.empty(self.len, dtype=np.float32)
        _lib.check(self._lib.aa_fft_inverse(self._plan, _lib.f32_ptr(s.view(np.float32)),
                                            _lib.f32_ptr(out)))
        return out

    # ---- batched, device-resident forms (torch CUDA tensors) ------------------
    def forward_batch(self, frames, out=None, stream=None):
        """frames: torch.float32 CUDA [batch, len] -> torch.complex64 [batch, len/2+1]."""
        import torch

        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous()
        batch = frames.shape[0]
        if out is None:
            out = torch.empty((batch, self.len + 2), dtype=torch.float32, device=frames.device)
        st = stream if stream is not None else torch.cuda.current_stream(frames.device).cuda_stream
        _lib.check(self._lib.aa_fft_forward_batch_device(self._plan, frames.data_ptr(),
                                                         out.data_ptr(), batch, st))
        return torch.view_as_complex(out.view(batch, self.len // 2 + 1, 2))

    def inverse_batch(self, spec, out=None, stream=None):
        """spec: torch.complex64 CUDA [batch, len/2+1] -> torch.float32 [batch, len]."""
        import torch

        assert spec.is_cuda and spec.dtype == torch.complex64 and spec.is_contiguous()
        batch = spec.shape[0]
        if out is None:
            out = torch.empty((batch, self.len), dtype=torch.float32, device=spec.device)
        st = stream if stream is not None else torch.cuda.current_stream(spec.device).cuda_stream
        _lib.check(self._lib.aa_fft_inverse_batch_device(
            self._plan, torch.view_as_real(spec).data_ptr(), out.data_ptr(), batch, st))
        return out

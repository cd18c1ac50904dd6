// src/dsp/fft.rs of audio-analyzer-rs with libaa_stft.so underneath: the same public surface as the reference's
// `FftProcessor` (reference: src/dsp/fft.rs:6-42 - `new(len)`, `process_forward(&mut [f32]) -> &[Complex<f32>]`,
// `process_inverse(&mut [Complex<f32>]) -> &[f32]`; callers: src/audio_io/stft.rs:180, 306 and
// src/analysis/onset.rs:141, 259), plus the batched plan the GPU is for.  Source only: this image has no rustc;
// tests/test_rust_binding.py checks every `aa_*` call below against include/aa_stft.h (name and argument count).
//
// Put aa_stft_sys.rs next to it as `mod aa_sys;` and link with
//     cargo:rustc-link-search=<dir of libaa_stft.so>   cargo:rustc-link-lib=dylib=aa_stft      (build.rs)
use super::aa_sys::*;
use rustfft::num_complex::Complex;

fn check(rc: i32) {
    if rc != AA_OK {
        // the reference panics here (`.unwrap()` on realfft's Result, fft.rs:69, 99)
        let msg = unsafe { std::ffi::CStr::from_ptr(aa_last_error()) }.to_string_lossy();
        panic!("libaa_stft: status {rc}: {msg}");
    }
}

/// `FftProcessor` (fft.rs:6-10): owns the plan and the two buffers its methods return borrows of (fft.rs:18-23).
pub struct FftProcessor {
    plan: *mut aa_fft_plan,
    len: usize,
    spectrum: Vec<Complex<f32>>,
    output: Vec<f32>,
}
// one processor per worker thread, constructed inside the thread closure (stft.rs:176-180, onset.rs:138-141)
unsafe impl Send for FftProcessor {}

impl FftProcessor {
    /// fft.rs:14-29
    pub fn new(len: usize) -> Self {
        let mut plan = std::ptr::null_mut();
        check(unsafe { aa_fft_create(len as i32, &mut plan) });
        Self { plan, len, spectrum: vec![Complex::default(); len / 2 + 1], output: vec![0.0; len] }
    }

    /// fft.rs:33-35 - unnormalised R2C; the returned borrow is valid until the next call.  (The reference uses the
    /// input as scratch; here it is left untouched.)
    pub fn process_forward(&mut self, windowed: &mut [f32]) -> &[Complex<f32>] {
        assert_eq!(windowed.len(), self.len);
        // Complex<f32> is #[repr(C)] {re, im}: the interleaved float pairs of the C ABI
        check(unsafe { aa_fft_forward(self.plan, windowed.as_ptr(), self.spectrum.as_mut_ptr() as *mut f32) });
        &self.spectrum
    }

    /// fft.rs:39-41 - unnormalised C2R (inverse(forward(x)) == len * x).
    pub fn process_inverse(&mut self, spec: &mut [Complex<f32>]) -> &[f32] {
        assert_eq!(spec.len(), self.len / 2 + 1);
        check(unsafe { aa_fft_inverse(self.plan, spec.as_ptr() as *const f32, self.output.as_mut_ptr()) });
        &self.output
    }

    pub fn len(&self) -> usize {
        unsafe { aa_fft_len(self.plan) as usize }
    }
}

impl Drop for FftProcessor {
    fn drop(&mut self) {
        unsafe { aa_fft_destroy(self.plan) };
    }
}

/// `STFT::hann_window` / `OnsetDetector::hann_window` (stft.rs:641-648, onset.rs:549-556), all-f32 as there.
pub fn hann_window(n: usize) -> Vec<f32> {
    let mut w = vec![0f32; n];
    check(unsafe { aa_hann_window(n as i32, w.as_mut_ptr()) });
    w
}

/// The frame loop of stft.rs:273-318 (window, FFT, `c.norm()`) over a whole recording in one call: row t of the
/// result is `magnitudes` of frame t (stft.rs:314-318); frame t = samples[t * hop .. t * hop + n_fft).
pub struct BatchedStft {
    plan: *mut aa_stft_plan,
    cols: usize,
}
unsafe impl Send for BatchedStft {}

impl BatchedStft {
    /// `output` is one of AA_OUT_*; 2048 / 512 is the tuner (stft.rs:169-170), 256 / 64 the onset detector
    /// (onset.rs:122-123).
    pub fn new(n_fft: usize, hop: usize, output: i32, sample_rate: i32, n_mels: i32, n_mfcc: i32) -> Self {
        let cfg = aa_stft_config {
            n_fft: n_fft as i32,
            hop: hop as i32,
            window: AA_WINDOW_HANN_PERIODIC,
            output,
            sample_rate,
            n_mels,
            n_mfcc,
            fmin: 0.0,
            fmax: 0.0,
            log_floor: 0.0,
        };
        let mut plan = std::ptr::null_mut();
        check(unsafe { aa_stft_plan_create(&cfg, std::ptr::null(), &mut plan) });
        let cols = unsafe { aa_stft_out_cols(plan) } as usize;
        Self { plan, cols }
    }

    pub fn num_frames(&self, n_samples: usize) -> usize {
        unsafe { aa_stft_num_frames(self.plan, n_samples as i64) as usize }
    }

    /// Host buffers in, host rows out (`frames x cols`); page-lock long-lived buffers with `aa_host_register` to run
    /// at the PCIe rate.
    pub fn execute(&mut self, samples: &[f32]) -> Vec<f32> {
        let mut rows = vec![0f32; self.num_frames(samples.len()) * self.cols];
        check(unsafe { aa_stft_execute_host(self.plan, samples.as_ptr(), samples.len() as i64, rows.as_mut_ptr()) });
        rows
    }

    /// Device pointers (hipMalloc'ed by the caller), frames [f0, f1) of the buffer, on `stream`.
    pub unsafe fn execute_device(&mut self, samples_dev: *const f32, n_samples: usize, f0: usize, f1: usize,
                                 out_dev: *mut f32, stream: *mut std::os::raw::c_void) {
        check(aa_stft_execute_device(self.plan, samples_dev, n_samples as i64, f0 as i64, f1 as i64, out_dev, stream));
    }
}

impl Drop for BatchedStft {
    fn drop(&mut self) {
        unsafe { aa_stft_plan_destroy(self.plan) };
    }
}

/// The live path: `aa_stream_push(slot)` replaces the body of the `while let Ok(idx) = cons.pop()` /
/// `while available_samples >= window_size` pair (stft.rs:240-318): same ring (`8192.max(4 * n_fft)`, stft.rs:171),
/// same drop-oldest rule (:262-266); one magnitude row and the absolute first-sample index per completed frame.
pub struct LiveStft {
    s: *mut aa_stream,
    cols: usize,
}
unsafe impl Send for LiveStft {}

impl LiveStft {
    pub fn new(n_fft: usize, hop: usize, ring_len: usize) -> Self {
        let cfg = aa_stft_config {
            n_fft: n_fft as i32,
            hop: hop as i32,
            window: AA_WINDOW_HANN_PERIODIC,
            output: AA_OUT_MAG,
            sample_rate: 0,
            n_mels: 0,
            n_mfcc: 0,
            fmin: 0.0,
            fmax: 0.0,
            log_floor: 0.0,
        };
        let mut s = std::ptr::null_mut();
        check(unsafe { aa_stream_create(&cfg, ring_len as i32, &mut s) });
        Self { s, cols: n_fft / 2 + 1 }
    }

    /// Returns the rows (`frames x cols`) and the frames' first-sample indices completed by this slot.
    pub fn push(&mut self, slot: &[f32], max_frames: usize) -> (Vec<f32>, Vec<i64>) {
        let mut rows = vec![0f32; max_frames * self.cols];
        let mut starts = vec![0i64; max_frames];
        let mut n: i64 = 0;
        check(unsafe {
            aa_stream_push(self.s, slot.as_ptr(), slot.len() as i32, rows.as_mut_ptr(), starts.as_mut_ptr(),
                           max_frames as i64, &mut n)
        });
        rows.truncate(n as usize * self.cols);
        starts.truncate(n as usize);
        (rows, starts)
    }
}

impl Drop for LiveStft {
    fn drop(&mut self) {
        unsafe { aa_stream_destroy(self.s) };
    }
}

/*
 * aa_stft.h — C ABI of the MI355X-native STFT / spectrogram hot path.
 *
 * Drop-in boundary for the one data-parallel path of audio-analyzer-rs:
 *   window -> real FFT -> |X|      (reference: src/dsp/fft.rs, the frame loops of
 *   src/audio_io/stft.rs:273-318 and src/analysis/onset.rs:244-272).
 *
 * Every entry point is extern "C", takes plain pointers and sizes, returns an
 * int status (AA_OK == 0, negative on error) and never aborts.  The reference
 * panics instead (`.unwrap()` at src/dsp/fft.rs:69, :99); a Rust wrapper can turn
 * a non-zero status back into a panic to keep that behaviour.
 *
 * Layout conventions
 *   - complex bins are interleaved (re, im) float pairs == num_complex::Complex<f32>
 *     (#[repr(C)]), n_fft/2+1 pairs per frame;
 *   - a batched output is row-major [frame][column], frames in stream order:
 *     frame t covers samples [t*hop, t*hop + n_fft)  (stft.rs:273, 296-299, 436-437);
 *   - "device" pointers are HIP device pointers on the plan's device; "host"
 *     pointers are ordinary host memory (staged through the plan's workspace);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *
 * Threading: a plan is not thread-safe; distinct plans are independent
 * (the reference builds one FftProcessor per worker thread: stft.rs:176-180,
 * onset.rs:138-141).  A plan binds to the HIP device current at creation.
 *
 * The library has NO CPU fallback: if no HIP device is usable every compute
 * entry point returns AA_ERR_HIP.
 */
#ifndef AA_STFT_H
#define AA_STFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AA_ABI_VERSION 4

/* ---- status codes -------------------------------------------------------- */
enum {
    AA_OK = 0,
    AA_ERR_ARG = -1,         /* NULL pointer / bad enum / bad range                 */
    AA_ERR_LEN = -2,         /* length the plan cannot transform (reference: panic
                                from realfft's Result, src/dsp/fft.rs:69)           */
    AA_ERR_HIP = -3,         /* HIP runtime error or no device                      */
    AA_ERR_NOMEM = -4,
    AA_ERR_UNSUPPORTED = -5, /* valid request this build does not implement         */
    AA_ERR_NCCL = -6         /* RCCL error, or librccl could not be bound (aa_comm_*) */
};

/* Last error text of the calling thread ("" if none). Never NULL. */
const char *aa_last_error(void);
int aa_abi_version(void);
/* Number of visible HIP devices (0 and AA_ERR_HIP if the runtime is unusable). */
int aa_device_count(int *count);

/* ---- window ------------------------------------------------------------- */
enum {
    AA_WINDOW_HANN_PERIODIC = 0, /* STFT::hann_window, stft.rs:641-648 / onset.rs:549-556 */
    AA_WINDOW_RECT = 1,          /* all ones (FftProcessor used bare)                       */
    AA_WINDOW_CUSTOM = 2         /* caller-supplied n_fft floats                            */
};

/* Periodic Hann exactly as the reference forms it, every step in f32:
 *   x = i as f32 / n as f32;  w = 0.5 - 0.5 * cos(2.0 * PI_f32 * x)
 * Replaces STFT::hann_window(n) (src/audio_io/stft.rs:641-648) and
 * OnsetDetector::hann_window(n) (src/analysis/onset.rs:549-556).  Host-side, tiny. */
int aa_hann_window(int n, float *out_host);

/* ---- FftProcessor mirror (src/dsp/fft.rs:6-42) --------------------------- */
typedef struct aa_fft_plan aa_fft_plan;

/* FftProcessor::new(len) — src/dsp/fft.rs:14-29.  len must be even and >= 4;
 * powers of two run the LDS radix-2 kernels, other even lengths a direct DFT
 * kernel (len <= 8192). */
int aa_fft_create(int len, aa_fft_plan **out_plan);
int aa_fft_destroy(aa_fft_plan *plan);
int aa_fft_len(const aa_fft_plan *plan);

/* FftProcessor::process_forward(&mut [f32]) -> &[Complex<f32>] — fft.rs:33-35, 66-71.
 * Unnormalised forward real DFT of ONE frame: in[len] -> out[2*(len/2+1)].
 * Unlike the reference the input is const (not used as scratch) and the output
 * buffer is caller-owned (the reference returns a borrow of plan-owned storage). */
int aa_fft_forward(aa_fft_plan *plan, const float *in_host, float *out_host);

/* FftProcessor::process_inverse(&mut [Complex<f32>]) -> &[f32] — fft.rs:39-41, 96-101.
 * Unnormalised inverse (inverse(forward(x)) == len * x): spec[2*(len/2+1)] -> out[len].
 * Imaginary parts of bin 0 and bin len/2 are ignored (status stays AA_OK). */
int aa_fft_inverse(aa_fft_plan *plan, const float *spec_host, float *out_host);

/* Batched forms on device memory: `batch` independent frames, contiguous rows
 * (row stride len floats in / 2*(len/2+1) floats out for forward, the reverse for
 * inverse).  These are what a GPU caller should use instead of per-frame calls. */
int aa_fft_forward_batch_device(aa_fft_plan *plan, const float *in_dev, float *out_dev,
                                int64_t batch, void *stream);
int aa_fft_inverse_batch_device(aa_fft_plan *plan, const float *spec_dev, float *out_dev,
                                int64_t batch, void *stream);

/* ---- batched STFT (the frame loops of stft.rs:273-318 / onset.rs:244-272) -- */
enum {
    AA_OUT_COMPLEX = 0,  /* Complex<f32> bins, 2*(n_fft/2+1) floats per frame (fft.rs:33)  */
    AA_OUT_MAG = 1,      /* |X| = c.norm(), n_fft/2+1 floats (stft.rs:314-318, onset.rs:271) */
    AA_OUT_POWER = 2,    /* |X|^2                         — build-defined, not in reference  */
    AA_OUT_LOGPOWER = 3, /* 10*log10(max(|X|^2, floor))   — build-defined, not in reference  */
    AA_OUT_LOGMEL = 4,   /* 10*log10(max(mel @ |X|^2, floor)), n_mels floats — build-defined:
                            Slaney mel scale, Slaney-normalised triangles, fmin..fmax, on the
                            f32 matrix cores (oracle/truth.py holds the float64 definition)    */
    AA_OUT_MFCC = 5      /* DCT-II(ortho) of log-mel, n_mfcc floats      — build-defined     */
};

/* Range contract of AA_OUT_MAG.  The reference's `c.norm()` is hypot(re, im) (stft.rs:314-318); the
 * kernels compute sqrt(re*re + im*im) in f32 (one v_sqrt_f32 per bin — hypot's rescaling would cost
 * the bandwidth-bound kernels ~5 % for inputs the reference's own pipeline cannot produce: its samples
 * are clamped to |x| <= 0.97, dynamics.rs:319-330, so |X| <= 0.97 * n_fft / 2).  Measured on gfx950
 * (tools/probe_mag_range.py, tests/test_gpu_parity.py::test_mag_edge_values):
 *   2^-63 (1.1e-19) <= |X| <= 2^63.9 (1.8e19)  within 2 ulp of hypotf(re, im) of the same f32 bin;
 *   |X| > 1.845e19 (re*re + im*im overflows)     +inf, where hypot returns the finite value;
 *   |X| < 2^-63 (the sum of squares is subnormal) 0 (flushed), where hypot returns the tiny value;
 *                                                 subnormal INPUT samples transform to 0 as well;
 *   a frame that holds a NaN or an Inf sample     every bin of THAT frame is NaN or Inf — as in the
 *                                                 reference, where the FFT spreads it to all bins —
 *                                                 but a bin with (re, im) = (inf, NaN) reads NaN here
 *                                                 and +inf from hypot; other frames are unaffected.
 * AA_OUT_COMPLEX (process_forward's own output, fft.rs:33-35) has no such limits: callers that need
 * hypot semantics outside the range above take the complex rows and apply hypotf themselves. */
typedef struct aa_stft_config {
    int32_t n_fft;       /* window / transform length: 2048 tuner (stft.rs:170), 256 onset
                            (onset.rs:122); BASELINE bench: 1024                           */
    int32_t hop;         /* hop size: n_fft/4 in the reference (stft.rs:169, onset.rs:123) */
    int32_t window;      /* AA_WINDOW_*                                                    */
    int32_t output;      /* AA_OUT_*                                                       */
    int32_t sample_rate; /* Hz; only used by mel outputs (reference streams run at the
                            device rate, e.g. 48000)                                       */
    int32_t n_mels;      /* AA_OUT_LOGMEL / AA_OUT_MFCC                                    */
    int32_t n_mfcc;      /* AA_OUT_MFCC                                                    */
    float fmin;          /* mel lower edge (Hz)                                            */
    float fmax;          /* mel upper edge (Hz); <= 0 means sample_rate/2                  */
    float log_floor;     /* clamp before log10; <= 0 means 1e-10                           */
} aa_stft_config;

typedef struct aa_stft_plan aa_stft_plan;

/* custom_window_host: n_fft floats when cfg->window == AA_WINDOW_CUSTOM, else NULL. */
int aa_stft_plan_create(const aa_stft_config *cfg, const float *custom_window_host,
                        aa_stft_plan **out_plan);
int aa_stft_plan_destroy(aa_stft_plan *plan);

/* floor((n_samples - n_fft)/hop) + 1, or 0 when n_samples < n_fft: the count the
 * `while available_samples >= window_size` loop emits for a gap-free stream
 * (stft.rs:273, 436-437).  Pure integer; bit-exact by construction. */
int64_t aa_stft_num_frames(const aa_stft_plan *plan, int64_t n_samples);
/* floats per output row for the plan's output kind. */
int32_t aa_stft_out_cols(const aa_stft_plan *plan);
/* Name of the FFT kernel family the plan selected: "wave1024", "wave2048" (one wave per frame), "wave512",
 * "wave256", "wave128" (several frames per wave), "wg_stockham" (one workgroup per frame: 4096 / 8192, and 1024 /
 * 2048 at hops that are not multiples of 4), "lds_stockham" (other powers of two), "dft" (even non-powers of two).
 * The wave kernels stage 16-byte pieces: a call whose first sample is not 16-byte aligned runs on the generic
 * kernel of the same length instead (same rows to rounding). */
const char *aa_stft_kernel_name(const aa_stft_plan *plan);
/* How an AA_OUT_LOGMEL / AA_OUT_MFCC plan reduces the power rows ("" for the other output kinds); it follows from
 * the plan's parameters alone:
 *   "mfma"         inside the wave kernel, the filterbank band-blocked on v_mfma_f32_4x4x1 (stft_wave1024_mel_kernel
 *                  at 8 frames per chunk, stft_wave2048_mel_kernel at hop 512; n_mels % 4 == 0, 16-byte-aligned rows);
 *   "mfma+dct"     MFCC: the same with the DCT-II fused behind it (n_fft 1024, n_mfcc <= 64) - one launch;
 *   "mfma+gemm"    MFCC at n_fft 2048, hop 512: log-mel slab from "mfma", then the DCT-II as an f32 MFMA GEMM;
 *   "gather"       inside the wave kernel on the VALU, weights rebuilt from the triangle edges (any band count / hop);
 *   "gather+gemm"  MFCC: log-mel slab from "gather", then the DCT-II as an f32 MFMA GEMM;
 *   "gemm"         power slab -> f32 MFMA filterbank GEMM (-> DCT GEMM): the generic FFT kernels, and any call whose
 *                  first sample is not 16-byte aligned. */
const char *aa_stft_mel_path_name(const aa_stft_plan *plan);

/* Frames [frame_begin, frame_end) of the buffer `samples_dev[0 .. n_samples)`,
 * rows written from out_dev[0].  frame indices are relative to samples_dev[0]:
 * frame t reads samples_dev[t*hop .. t*hop+n_fft).  Asynchronous on `stream`. */
int aa_stft_execute_device(aa_stft_plan *plan, const float *samples_dev, int64_t n_samples,
                           int64_t frame_begin, int64_t frame_end, float *out_dev,
                           void *stream);
/* n_channels equal-length mono channels (planar: channel c is samples_dev[c * channel_stride ..
 * + n_samples)), every frame of each, rows of channel c from out_dev[c * out_channel_stride]
 * (floats; out_channel_stride >= aa_stft_num_frames * aa_stft_out_cols).  The reference never runs
 * a multi-channel STFT (it averages stereo to mono at capture, mod.rs:764, 789-793): this is the
 * "two independent mono channels" form of BASELINE configs[2].  Row c * frames + t is bit-identical
 * to what aa_stft_execute_device writes for channel c alone; with the wave kernels of n_fft 1024 /
 * 2048 and both strides multiples of 4 floats all channels run in ONE launch (the persistent
 * workgroups are split evenly between them), otherwise channel after channel.  Asynchronous. */
int aa_stft_execute_channels_device(aa_stft_plan *plan, const float *samples_dev, int64_t n_samples,
                                    int64_t channel_stride, int32_t n_channels, float *out_dev,
                                    int64_t out_channel_stride, void *stream);
/* All frames of a host buffer; synchronous (H2D, kernels, D2H through the
 * plan's workspace).  out_host holds aa_stft_num_frames * aa_stft_out_cols floats. */
int aa_stft_execute_host(aa_stft_plan *plan, const float *samples_host, int64_t n_samples,
                         float *out_host);

/* A row buffer for `plan`, n_frames * aa_stft_out_cols floats of device memory, PLACED BY MEASUREMENT.  Why:
 * process_forward's own rows (AA_OUT_COMPLEX, fft.rs:33-35) stream out of the 1024 / 256 launch at ~6.4 TB/s into some
 * buffers and ~5.1 TB/s into others (89-95 us against 118 us for ten minutes of audio) - a property of the PHYSICAL
 * memory behind the buffer that lasts the buffer's lifetime: about one 462-MB slab in twenty of a device is fast, in
 * runs of neighbours; same write requests, the DRAM side returns write credits 1.4 x more slowly in the others
 * (DESIGN.md section 5).  No virtual address or kernel parameter reaches that, so: slab after slab is allocated (all
 * held meanwhile, so that each is other physical memory) and the plan's own launch timed into it (a few launches on
 * the default stream, on scratch samples) until one is clearly faster than the median of those seen (re-timed to
 * confirm) or max_candidates slabs / the device's free memory less 2 GiB are used up - then the fastest seen is kept;
 * the rest is freed.  The magnitude rows do not take part (every slab within 2 %): for them, or with max_candidates
 * = 1, this is a plain allocation.  kept_us / typical_us (nullable): the launch time into the kept slab and the median
 * over the slabs tried.  Synchronous; typically 10-60 slabs and well under a second, transiently that many x the
 * buffer.  Release with aa_rows_free. */
int aa_rows_alloc(aa_stft_plan *plan, int64_t n_frames, int32_t max_candidates, float **out_rows_dev, float *kept_us,
                  float *typical_us);
int aa_rows_free(float *rows_dev);

/* Page-lock / release a caller-owned host buffer (hipHostRegister).  With BOTH buffers of
 * aa_stft_execute_host registered, the call streams frame chunks over two HIP streams (H2D,
 * kernel, D2H overlapped) instead of staging through pageable copies.  The caller must
 * unregister before freeing the memory. */
int aa_host_register(void *ptr, size_t bytes);
int aa_host_unregister(void *ptr);

/* ---- capture-side downmix (src/audio_io/mod.rs:784-794) -------------------- */
/* mono[i] = (sum of the first min(channels, 2) channels of frame i) / that count: what the
 * input callback does before a slot reaches the STFT workers (the reference never runs a
 * multi-channel STFT).  interleaved_dev holds n_frames * channels floats. */
int aa_downmix_device(const float *interleaved_dev, int64_t n_frames, int32_t channels,
                      float *mono_dev, void *stream);

/* ---- onset front end: spectral flux + frame energy (src/analysis/onset.rs:261-291) ---- */
/* First stage after the onset STFT: for every frame t of a magnitude
 * spectrogram mags_dev[n_frames][n_bins] (AA_OUT_MAG rows; the reference geometry is 256/64,
 * n_bins = 129; n_bins <= 4160)
 *   energy[t] = sum_i mag[t][i]
 *   flux[t]   = sum_i max(0, smooth3(mag[t])[i] - mag[t-1][i]) * (1 - i/n_bins)
 * with smooth3 the 3-bin mean (ends unsmoothed) and mag[-1] = prev_row_dev (NULL = zeros, the
 * detector's initial state, onset.rs:146; a shard passes its predecessor's last row).
 * ABI version 4: one lane per frame walks the bins in order, i.e. both sums are the reference's own sequential
 * f32 sums (no contraction, the weight and the division by 3 rounded as there): flux and energy are BIT-IDENTICAL to
 * the sequential evaluation, for any split of the frames into calls or shards.  (Versions 1-3 reduced a frame
 * across a wave: 1e-5.)  The rows of a block of 64 frames are fetched once, as one contiguous span. */
int aa_onset_flux_device(const float *mags_dev, int64_t n_frames, int32_t n_bins,
                         const float *prev_row_dev, float *flux_dev, float *energy_dev, void *stream);

/* ---- onset detector: burst floor, gates and trackers (src/analysis/onset.rs:298-371, 531-535) ---- */
/* The per-frame logic that follows the flux in the reference's onset thread ("next" row f-2), on the device:
 *   per bin   r = mag / max(floor, max(global_floor, 0.01));  r > 2.5: burst, floor = 1.3 mag;
 *             else floor += (mag > floor ? 0.1 : 0.04) * (mag - floor)            (onset.rs:311-332)
 *   frame     burst_count, max_excess = max r; flux = 0 when burst_count < 2       (:337-339)
 *             energy EMA (0.84 rising / 0.95 falling)                              (:345-350)
 *             FluxTracker(1.5, 0.84, 0.89)::update -> flux_onset                   (:66-83, :158)
 *             bin_burst_onset = max_excess > 3 && burst_count >= 3; onset_detected (:355-357)
 *             energy_rising = energy > 1.5 ema (:373); velocity (:389-390)
 *             gated = onset_detected && !suppressed_by_tick && energy_rising && frames_since_onset >= 3   (:402)
 *             frames_since_onset = 0 when the reference sets onset_fired, or at a detected frame within 3 frames of
 *             the last one; + 1 otherwise                                          (:535-539)
 * What the decision takes from the MusicalTransport is supplied by the caller (it does not come from the spectrum):
 *   frame_mask_dev  one byte per frame (NULL = 0, under AA_ONSET_CALIBRATING = AA_ONSET_MASK_CAL_OPEN):
 *       AA_ONSET_MASK_TICK         the nearest metronome tick is inside the tick guard (tick_dist < TICK_GUARD_S * bpm
 *                                  / 60, :393-396): a detected onset there is dropped (flags bit 5)
 *       AA_ONSET_MASK_CAL_OPEN     the calibration click has been played and an onset stamped at this frame would leave
 *                                  a plausible residual, 0 .. 0.5 s behind it (:405-422: target != 0, not rejected)
 *       AA_ONSET_MASK_CAL_TIMEOUT  the calibration timeout has run out by this frame (:361-371)
 *   calibrated
 *       AA_ONSET_UNCALIBRATED  the calibration is running and its click has not been played: a gated onset is
 *                              "pre-calibration noise" (:406-408) - nothing fires, frames_since_onset counts on
 *       AA_ONSET_CALIBRATED    calibration_done: every gated onset is an event (flags bit 4)
 *       AA_ONSET_CALIBRATING   calibration_done is CARRIED IN THE STATE (0 in a fresh detector): a frame with
 *                              CAL_TIMEOUT sets it in front of its decision (:361-371); a gated onset on a frame with
 *                              CAL_OPEN sets it and is CONSUMED by the calibration (flags bit 7, no event) - the
 *                              reference sets onset_fired there without pushing the event (:424-428), so
 *                              frames_since_onset restarts and the stream goes on calibrated; a gated onset on any other
 *                              frame is ignored as under AA_ONSET_UNCALIBRATED (:406-408, :417-422).  The caller reads
 *                              the residual off the frame index of the bit-7 frame (it owns the transport).
 * (ABI version 3 had "non-zero = tick" and a 0 / 1 flag, under which a stream could not cross its own calibration
 * click: the consuming onset must restart frames_since_onset, and whether it is gated depends on that counter.)
 * Inputs: magnitude rows [n_frames x n_bins], flux / frame energy per frame (aa_onset_flux_device),
 * global_floor per frame (= 10^(noise_floor_db/20) * n_bins / 2, onset.rs:299-300).
 * state_dev: n_bins + 4 floats {floors, energy_ema, tracker threshold, frames_since_onset (int bits),
 * calibration_done (int bits)}, updated in place; `initialized` = 0 starts a fresh detector.  A call with
 * n_frames == 0 leaves the state untouched (a fresh detector stays fresh: pass initialized = 0 again).
 * Outputs per frame: gated flux, energy EMA, max_excess, burst count, velocity and
 * flags = flux_onset | bin_burst_onset << 1 | onset_detected << 2 | energy_rising << 3 | event << 4 |
 * suppressed_by_tick << 5 | gated << 6 | consumed_by_calibration << 7.  Every f32 operation is the reference's, in its
 * order: all outputs are bit-identical to the sequential evaluation on the same inputs.  Up to 192 bins (the onset
 * geometry has 129) a stream runs as one workgroup of five waves with the recurrences taken apart - the per-bin floors
 * one lane per bin, the burst count / largest excess of a frame as a column walk, the scalar chains in a wave of their
 * own, a block of 32 frames apart (a ten-minute stream: ~30 ms; one wave walking the frames took 313 ms) - at any
 * number of streams (aa_onset_gates_batch_device); wider spectra keep the one-wave form. */
enum { AA_ONSET_MASK_TICK = 1, AA_ONSET_MASK_CAL_OPEN = 2, AA_ONSET_MASK_CAL_TIMEOUT = 4 };
enum { AA_ONSET_UNCALIBRATED = 0, AA_ONSET_CALIBRATED = 1, AA_ONSET_CALIBRATING = 2 };
enum {
    AA_ONSET_FLAG_FLUX = 1, AA_ONSET_FLAG_BURST = 2, AA_ONSET_FLAG_DETECTED = 4, AA_ONSET_FLAG_ENERGY_RISING = 8,
    AA_ONSET_FLAG_EVENT = 16, AA_ONSET_FLAG_TICK = 32, AA_ONSET_FLAG_GATED = 64, AA_ONSET_FLAG_CAL_CONSUMED = 128
};
int aa_onset_gates_device(const float *mags_dev, int64_t n_frames, int32_t n_bins, const float *flux_dev,
                          const float *energy_dev, const float *global_floor_dev, const uint8_t *frame_mask_dev,
                          int32_t calibrated, float *state_dev, int32_t initialized, float *out_flux_dev,
                          float *out_ema_dev, float *out_excess_dev, int32_t *out_count_dev, uint8_t *out_flags_dev,
                          float *out_velocity_dev, void *stream);

/* ---- tuner front end: per-bin adaptive noise floor (src/audio_io/stft.rs:326-367) ---- */
/* First stage after the tuner STFT.  For every bin a first-order
 * recurrence over frames with the reference's constants (stft.rs:219-224):
 *   first frame:  floor = max(mag, 5*global_floor); prev = mag
 *   later:        vol = 0.75*vol + 0.25*|mag - prev|; prev = mag;
 *                 sustained = mag/max(floor,0.01) > 1.5 && clamp(vol/max(mag,0.05),0,1) < 0.15;
 *                 if !sustained: floor += (mag > floor ? 0.04 + 0.31*vol_norm : 0.02)*(mag - floor)
 *   eff_floor[t][k] = min(floor[k], 2.5*global_floor[t])
 * global_floor_dev[t] = 10^(noise_floor_db/20) * n_bins / 2 (stft.rs:322-324) comes from the
 * caller.  state_dev holds 3*n_bins floats {floor, prev_mag, volatility} and is updated in
 * place, `initialized` says whether it carries a previous call's state — so a stream can be fed
 * in pieces.  One lane per bin walks the frames in order with the reference's operation order:
 * results are bit-identical to the sequential f32 evaluation.  The pitch extraction and
 * PitchTracker that consume eff_floor (stft.rs:443-620, 28-117) stay on the host. */
int aa_tuner_floor_device(const float *mags_dev, int64_t n_frames, int32_t n_bins,
                          const float *global_floor_dev, float *state_dev, int32_t initialized,
                          float *eff_floor_dev, void *stream);

/* ---- tuner: multi-pitch extraction per frame (src/audio_io/stft.rs:443-620) ---- */
/* STFT::extract_pitches for every frame of a magnitude spectrogram and its effective noise floor
 * (aa_tuner_floor_device): peaks above the floor in [min_freq, max_freq] (the reference passes
 * 24 Hz and 10 kHz, stft.rs:173-174; bin_width = sample_rate / n_fft, :320), parabolic
 * log-magnitude interpolation, harmonic-sum score over up to 14 harmonics, 50 % score cutoff,
 * harmonic-ghost suppression, descending score, 2-bin dedupe, at most 8 notes.
 * out_freq_dev / out_score_dev: [n_frames][8] (unused slots 0), out_count_dev: notes per frame;
 * n_bins <= 1537 (the rows of one frame and its peak lists live in one wave's LDS).
 * f32 in the reference's operation order; ln / log2 come from the device math library, so scores
 * match a CPU evaluation to ~1e-6 relative and a decision can only flip on an exact tie; equal
 * scores keep ascending bin order (the reference's sort_unstable_by leaves it unspecified).
 * The PitchTracker hysteresis that follows (stft.rs:28-117) is sequential host logic. */
int aa_tuner_pitches_device(const float *mags_dev, const float *eff_floor_dev, int64_t n_frames,
                            int32_t n_bins, float bin_width, float min_freq, float max_freq,
                            float *out_freq_dev, float *out_score_dev, int32_t *out_count_dev, void *stream);

/* ---- batches of independent streams ------------------------------------------------ */
/* The detectors above are recurrences over the frames of ONE stream; on a GPU they pay off across
 * streams (recordings, channels).  The batch forms take streams concatenated along the frame axis:
 * frame_offsets_dev[n_streams + 1] (device, ascending, [0] = 0, [n_streams] = n_frames_total);
 * stream s owns frames [off[s], off[s+1]) of every per-frame array.  State is [n_streams] x the
 * single-stream state; initialized_dev[n_streams] (NULL = all fresh).  One workgroup of five waves (gates up
 * to 192 bins; one wave beyond) / one lane per bin (floor) per stream; results are identical to per-stream
 * calls.  A stream without frames keeps its state as it was (and stays uninitialised if it was).
 * aa_onset_flux_batch_device: prev_rows_dev [n_streams][n_bins] or NULL (zeros) is the row before
 * each stream's first frame.  aa_tuner_pitches_device needs no batch form (frames independent). */
int aa_onset_flux_batch_device(const float *mags_dev, const int64_t *frame_offsets_dev, int32_t n_streams,
                               int64_t n_frames_total, int32_t n_bins, const float *prev_rows_dev,
                               float *flux_dev, float *energy_dev, void *stream);
int aa_onset_gates_batch_device(const float *mags_dev, const int64_t *frame_offsets_dev, int32_t n_streams,
                                int64_t n_frames_total, int32_t n_bins, const float *flux_dev,
                                const float *energy_dev, const float *global_floor_dev, const uint8_t *frame_mask_dev,
                                int32_t calibrated, float *state_dev, const int32_t *initialized_dev,
                                float *out_flux_dev, float *out_ema_dev, float *out_excess_dev,
                                int32_t *out_count_dev, uint8_t *out_flags_dev, float *out_velocity_dev,
                                void *stream);
int aa_tuner_floor_batch_device(const float *mags_dev, const int64_t *frame_offsets_dev, int32_t n_streams,
                                int64_t n_frames_total, int32_t n_bins, const float *global_floor_dev,
                                float *state_dev, const int32_t *initialized_dev, float *eff_floor_dev,
                                void *stream);

/* ---- input conditioning of the reducer thread (audio_io/mod.rs:351-478, dynamics.rs:152-360), host ---- */
/* The step before the path ("next" row f-3): per 1024-sample slot a 40 Hz high-pass and a 14 kHz
 * low-pass biquad (Q 0.707), an envelope gate (-60 dB, 40 ms release, 20 ms hold, ratio^4) and the
 * DynamicsTracker (slot RMS; noise floor = 10th percentile of the quiet slots; active / broadband
 * classification; median and 95th percentile of the playing slots; AGC towards -18 dB with a
 * 240 s time constant and a 0.97 peak clamp; dynamic level).  Sample-sequential IIRs and per-slot
 * sorts: host work, as in the reference.  It produces the two things the detectors take from
 * outside the path: the conditioned samples and noise_floor_db per slot, from which
 * global_floor = 10^(noise_floor_db / 20) * n_bins / 2 (stft.rs:322-324, onset.rs:299-300).
 * samples_host: n_slots * slot_len samples, processed in place.  Per-slot outputs as the
 * reference's DynamicsOutput (dynamics.rs:340-347); level: 0 = Silence, 1..8 = Ppp..Fff. */
typedef struct aa_conditioner aa_conditioner;
int aa_conditioner_create(float sample_rate, int32_t slot_len, aa_conditioner **out);
int aa_conditioner_destroy(aa_conditioner *c);
int aa_conditioner_process(aa_conditioner *c, float *samples_host, int64_t n_slots, float *noise_floor_db,
                           float *rms_db, float *gain_db, float *session_median_db, int32_t *level);

/* ---- tuner: PitchTracker hysteresis (src/audio_io/stft.rs:28-117), host side ---- */
/* Sequential over frames and tiny (<= 8 raw pitches per frame): host logic as in the reference.
 * A raw pitch within 3 % of an unmatched track updates it (EMA 0.6 / 0.4, or a snap when the
 * frame carries an onset) and raises its life (max 3); unmatched tracks lose a life (all of it on
 * an onset); tracks with life >= 2 are emitted.  Inputs: the per-frame arrays of
 * aa_tuner_pitches_device copied to the host, onset_host[t] != 0 marks a pending onset (NULL =
 * none).  Outputs: [n_frames][AA_PITCH_TRACKS_MAX] stable (freq, score), unused slots 0. */
#define AA_PITCH_TRACKS_MAX 32
typedef struct aa_pitch_tracker aa_pitch_tracker;
int aa_pitch_tracker_create(aa_pitch_tracker **out);
int aa_pitch_tracker_destroy(aa_pitch_tracker *t);
int aa_pitch_tracker_process(aa_pitch_tracker *t, const float *freq_host, const float *score_host,
                             const int32_t *count_host, const uint8_t *onset_host, int64_t n_frames,
                             float *out_freq_host, float *out_score_host, int32_t *out_count_host);

/* ---- tuner: TunerOutput - note / interval naming and the polled JSON (host) ---- */
/* What the reference's tuner thread makes of each stable pitch list ("next" row f-4, the JSON side):
 * src/analysis/theory.rs (Note, MidiNote, Interval) and the body of Tuner::run
 * (src/analysis/tuner.rs:141-207); Tuner::poll_output (src/lib.rs:80-82) serialises the result with
 * serde_json.  A few scalar f32 operations per frame: host code, as in the reference.  This is the
 * one stage the reference pins with known-answer tests (theory.rs:404-627); tests/test_theory.py
 * restates them against these entry points.
 * base_freq is the A4 reference (the reference's `Option<f32>` None = 440.0). */
#define AA_NOTE_NAME_LEN 8
#define AA_TUNER_NOTES_MAX 32
enum { AA_TUNER_MULTI_PITCH = 0, AA_TUNER_SINGLE_PITCH = 1 };                         /* tuner.rs:20-25 */
enum { AA_TUNING_EQUAL_TEMPERAMENT = 0, AA_TUNING_JUST_INTONATION = 1, AA_TUNING_PYTHAGOREAN = 2 }; /* :12-18 */

/* Note::from_freq + get_name + get_cents (theory.rs:199-254): nearest equal-tempered pitch, sharps
 * only ("A4", "C#5"), and the offset from it in cents. */
int aa_note_from_freq(float freq, float base_freq, char name[AA_NOTE_NAME_LEN], float *cents);
/* Note::try_new + to_freq (theory.rs:104-197): letter, optional accidental (# b x B n), octave 0..255;
 * cents is the Note's offset (0 for a parsed name, get_cents() for one made by from_freq).
 * A name that does not parse is AA_ERR_ARG with the reason in aa_last_error (the reference: Err). */
int aa_note_to_freq(const char *name, float cents, float base_freq, float *freq);
/* MidiNote::from_freq / to_freq (theory.rs:16-28, 40-43); midi is a u8 there (0..255). */
int aa_midi_from_freq(float freq, float base_freq, int32_t *midi, float *cents);
int aa_midi_to_freq(int32_t midi, float cents, float base_freq, float *freq);
/* Interval::new + get_name + get_accuracy (theory.rs:289-381): freqs[1] / freqs[0] folded into
 * (0, 2], nearest of the 13 ratios of the tuning system -> "Per8" "Min2" ... "Maj7", accuracy =
 * -ln(nearest / ratio) * 1732.5.  n < 2 or freqs[0] == 0 gives "Per8" / 0 as there.  An infinite
 * ratio is AA_ERR_ARG (the reference's folding loop would not terminate). */
int aa_interval(const float *freqs, int32_t n, int32_t system, char name[AA_NOTE_NAME_LEN], float *accuracy);

/* `struct TunerOutput` (tuner.rs:38-56) without the key string, which the caller owns. */
typedef struct aa_tuner_output {
    char label[256];                                      /* "C#4", "Per5", or the note names joined by spaces */
    float cents;
    int32_t n_notes;                                      /* 0: empty pitch list - the reference keeps its previous output */
    char notes[AA_TUNER_NOTES_MAX][AA_NOTE_NAME_LEN];
    float accuracies[AA_TUNER_NOTES_MAX];
    int32_t mode, system;
    float base_freq;
    double beat_position;
} aa_tuner_output;
/* One pass of the tuner loop (tuner.rs:153-207) over one frame's stable pitches (a row of
 * aa_pitch_tracker_process's output, n = its count): one pitch or SinglePitch mode -> the note of
 * the highest score (the last of equal scores); two -> sorted, the interval names the label and
 * its accuracy the cents; three or more -> the note names joined.  base_freq is clamped to
 * [220, 880] as TunerCommand::SetBaseFreq does (tuner.rs:119). */
int aa_tuner_output_build(const float *freq_host, const float *score_host, int32_t n, int32_t mode,
                          int32_t system, float base_freq, double beat_position, aa_tuner_output *out);
/* The string Tuner::poll_output returns: {"label":..,"cents":..,"notes":[..],"accuracies":[..],
 * "mode":"MultiPitch","system":"EqualTemperament","base_freq":440.0,"key":"C major",
 * "beat_position":..}, floats in serde_json's shortest round-trip form.  *len receives the length
 * without the terminator; AA_ERR_LEN when cap is too small (len is still set). */
int aa_tuner_output_json(const aa_tuner_output *out, const char *key, char *buf, size_t cap, size_t *len);

/* ---- frame-range sharding (one rank per GPU) --------------- */
/* Rank r of `world` owns frames [floor(r*F/world), floor((r+1)*F/world)) and needs
 * samples [f0*hop, (f1-1)*hop + n_fft) — a halo of n_fft-hop samples shared with
 * the next rank.  Frame indices stay global, so indexing is bit-exact whatever the
 * world size.  Empty ranges give f0 == f1 and s0 == s1. */
int aa_shard_frames(int64_t total_frames, int32_t world, int32_t rank, int64_t *f0,
                    int64_t *f1);
int aa_shard_samples(int64_t f0, int64_t f1, int32_t n_fft, int32_t hop, int64_t *s0,
                     int64_t *s1);

/* ---- collectives of the sharded job (RCCL over xGMI, one process per GPU) -- */
/* The partition above needs no collective: every rank transforms its own frame range and keeps
 * its rows.  These helpers cover the two exchanges a host may still want: the
 * samples start on ONE GPU (scatter), or one GPU wants every row (gather).  They run on librccl
 * directly (grouped ncclSend / ncclRecv, unpadded blocks, asynchronous on `stream`), so a Rust or
 * C++ host needs nothing but this library.  librccl is bound at run time; without it every entry
 * point returns AA_ERR_NCCL.  The reference has no counterpart (it is a single-process CPU
 * engine); the seam these sit behind is the frame loop of stft.rs:273-318 run over a shard.
 *
 * Bootstrap as with NCCL: rank 0 calls aa_comm_unique_id and hands the 128 bytes to the other
 * ranks by any means it has (a file, a pipe, MPI, the launcher's store); then EVERY rank calls
 * aa_comm_create with the same id, its rank and the world size, after selecting its GPU
 * (hipSetDevice) — the communicator binds to the current device, and every later call that takes
 * the communicator selects that device for its own duration (AA_ERR_HIP if it cannot), so it may
 * come from any host thread.
 *
 * Verification status: at world size 1 these run on the real librccl on one MI355X
 * (tests/test_comm_gpu.py).  For worlds of 2, 3 and 8 ranks the send / receive lists every rank
 * issues — peers, offsets, counts, groups, the all-gather form, totals below the world size — are
 * pinned against aa_shard_frames / aa_shard_samples with a loopback stand-in for librccl
 * (tests/test_comm_stub.py); RCCL itself moving those bytes between GPUs has not run yet (no
 * multi-GPU node was available): multi-rank transport is UNVERIFIED on hardware. */
#define AA_COMM_ID_BYTES 128
typedef struct aa_comm_id {
    char bytes[AA_COMM_ID_BYTES];
} aa_comm_id;
typedef struct aa_comm aa_comm;

int aa_comm_unique_id(aa_comm_id *out_id);
int aa_comm_create(const aa_comm_id *id, int32_t world, int32_t rank, aa_comm **out_comm);
int aa_comm_destroy(aa_comm *comm);
int32_t aa_comm_world(const aa_comm *comm);
int32_t aa_comm_rank(const aa_comm *comm);
/* Which librccl was bound ("librccl.so.1 already mapped in this process", a path, ...), or the
 * reason none could be.  Never NULL. */
const char *aa_comm_backend(void);
/* Every rank exchanges one float with every rank (itself included) and waits for the stream:
 * returns when all ranks have entered. */
int aa_comm_barrier(aa_comm *comm, void *stream);

/* Rank `root` holds the whole buffer full_dev[0, n_samples) (ignored elsewhere); every rank
 * receives the range aa_shard_samples gives it — its frames plus the n_fft-hop halo — into
 * local_dev (capacity in floats; AA_ERR_LEN if too small).  Point-to-point, not a broadcast: the
 * root sends n_samples + (world-1)*(n_fft-hop) floats in all. */
int aa_scatter_samples(aa_comm *comm, const float *full_dev, int64_t n_samples, int32_t n_fft,
                       int32_t hop, int32_t root, float *local_dev, int64_t local_capacity,
                       void *stream);
/* Every rank contributes its rows_dev[f1-f0][cols] (the frame range aa_shard_frames gives it);
 * rank `root` receives the dense [total_frames][cols] slab in out_dev (ignored elsewhere), each
 * block landing at its global row.  root == -1: every rank receives the slab (all-gather).
 * rows_dev may already be the rank's own slice of out_dev (no copy then). */
int aa_gather_rows(aa_comm *comm, const float *rows_dev, int64_t total_frames, int32_t cols,
                   int32_t root, float *out_dev, void *stream);

/* The same two exchanges cut into n_pieces pieces per rank, so that a host can overlap them with the compute:
 * piece k of the frame range [f0, f1) is [f0 + floor(k n / P), f0 + floor((k + 1) n / P)), n = f1 - f0 (the floor
 * rule of aa_shard_frames once more; pieces of short ranges may be empty).  Pure integer. */
int aa_shard_piece(int64_t f0, int64_t f1, int32_t n_pieces, int32_t piece, int64_t *p0, int64_t *p1);
/* Piece `piece` of aa_scatter_samples: the samples that piece ADDS to what the earlier pieces delivered (consecutive
 * pieces share n_fft - hop samples; every sample travels once), written at their place in local_dev - which holds
 * the rank's WHOLE range, as for aa_scatter_samples.  Pieces must be issued in order, the same on every rank. */
int aa_scatter_samples_piece(aa_comm *comm, const float *full_dev, int64_t n_samples, int32_t n_fft, int32_t hop,
                             int32_t root, float *local_dev, int64_t local_capacity, int32_t n_pieces,
                             int32_t piece, void *stream);
/* Piece `piece` of aa_gather_rows: rows_dev is the rank's WHOLE [f1 - f0][cols] buffer, the piece's rows are taken
 * from their place in it and land at their global rows of out_dev. */
int aa_gather_rows_piece(aa_comm *comm, const float *rows_dev, int64_t total_frames, int32_t cols, int32_t root,
                         float *out_dev, int32_t n_pieces, int32_t piece, void *stream);

/* End to end on rank `root`'s buffer with the three legs overlapped: scatter of piece k + 1 || compute of piece k ||
 * gather of piece k - 1.  The exchanges run on a stream owned by their communicator (comm_in: scatter, comm_out:
 * gather; pass the same communicator twice and the two kinds of exchange take turns on it, pass two and they run
 * side by side), the compute on `stream`; events order piece k's compute behind its samples and its gather behind
 * its rows.  For the caller the whole call is stream-ordered on `stream`: it starts behind what was queued there
 * and `stream` continues behind the last gather.  `compute(user, frame_begin, frame_end, stream)` must queue the
 * transform of local frames [frame_begin, frame_end) - relative to the rank's range, as local_dev and rows_dev are
 * - on `stream` and return AA_OK; it is called once per non-empty piece, in order.  One pipeline at a time per
 * communicator.  Every rank issues S0, S1, G0, S2, G1, ... in this order (what RCCL requires of one communicator).
 * Errors: a failure of the compute side (the callback, an event call) on one rank does NOT stop that rank's remaining
 * exchanges - its peers' matching sends / receives would otherwise wait for ever; the rows of the pieces behind the
 * failure are undefined and the first error is returned.  A failed exchange (AA_ERR_NCCL, or a HIP error inside one)
 * ends the posting; the communicators are unusable afterwards and must be destroyed.  Either way `stream` is joined
 * behind everything that was queued. */
typedef int (*aa_piece_fn)(void *user, int64_t frame_begin, int64_t frame_end, void *stream);
int aa_sharded_pipeline(aa_comm *comm_in, aa_comm *comm_out, const float *full_dev, int64_t n_samples,
                        int32_t n_fft, int32_t hop, int32_t cols, int32_t root, float *local_dev,
                        int64_t local_capacity, float *rows_dev, float *out_dev, int32_t n_pieces,
                        aa_piece_fn compute, void *user, void *stream);
/* aa_sharded_pipeline with aa_stft_execute_device of `plan` as the compute: rank `root`'s samples in, the dense
 * [total_frames][cols] slab on `root` out.  local_dev: the rank's sample range (capacity in floats), rows_dev its
 * [f1 - f0][cols] rows; on the root local_dev may be full_dev + s0 and rows_dev its slice of out_dev (no copies).
 * The plan, both communicators and `stream` belong to one device (AA_ERR_ARG if the plan's and the communicators'
 * differ); local_dev - and rows_dev of a log-mel / MFCC plan - must be 16-byte aligned (AA_ERR_ARG otherwise: from an
 * odd sample buffer the pieces would take the generic kernels and the rows would no longer be bit-equal to the
 * unsharded launch). */
int aa_stft_execute_sharded(aa_stft_plan *plan, aa_comm *comm_in, aa_comm *comm_out, const float *full_dev,
                            int64_t n_samples, int32_t root, float *local_dev, int64_t local_capacity,
                            float *rows_dev, float *out_dev, int32_t n_pieces, void *stream);

/* ---- diagnostics (no reference counterpart; used by the test suite and bench.py) ---- */
/* The flux kernel forms the reference's `/ 3.0` (onset.rs:268) in four instructions (x * RN(1/3), the exact remainder
 * by one FMA, one correcting FMA, v_div_fixup_f32).  Runs the device function over the `count` consecutive f32 bit
 * patterns from first_bits (count <= 2^32) and returns how many differ from the IEEE division (NaN == NaN):
 * tests/test_gpu_parity.py sweeps all 2^32 and expects 0.  Synchronous. */
int aa_diag_flux_div3_device(uint32_t first_bits, uint64_t count, uint64_t *mismatches_host);
/* What the memory system delivers with NO arithmetic, in the product's access idiom (16-byte loads, 16-byte nontemporal
 * stores, persistent workgroups on contiguous runs): writes_per_read = 1 copies n_floats floats, = 2 / 4 stores every
 * piece two / four times (dst_dev holds that many x n_floats: the 1 : 2 read : write mix of the magnitude launch, the
 * 1 : 4 mix of the complex rows).  n_floats % 4 == 0, both buffers 16-byte aligned.  bench.py times it next to the
 * headline kernel (roofline.copy_ceiling).  Asynchronous. */
int aa_diag_stream_copy_device(const float *src_dev, float *dst_dev, int64_t n_floats, int32_t writes_per_read,
                               void *stream);

/* ---- streaming front end (ring-buffer ingest of stft.rs:198-201, 240-266) -- */
typedef struct aa_stream aa_stream;

/* ring_len: 8192.max(4*n_fft) for the tuner (stft.rs:171), 4096 for onsets
 * (onset.rs:124); pass 0 for max(8192, 4*n_fft).  The stream owns a plan-shaped
 * GPU pipeline; frames are computed in batches when aa_stream_push is called. */
int aa_stream_create(const aa_stft_config *cfg, int32_t ring_len, aa_stream **out_stream);
int aa_stream_destroy(aa_stream *stream);
/* Append one slot of samples (the reference's slots are 1024 f32, mod.rs:126-128),
 * apply the overflow rule (oldest samples dropped, stft.rs:262-266), then emit every
 * complete frame: rows are written to out_host (capacity max_frames rows) and the
 * absolute stream index of each frame's first sample to frame_starts (may be NULL).
 * *n_emitted receives the number of rows written. */
int aa_stream_push(aa_stream *stream, const float *slot_host, int32_t slot_len,
                   float *out_host, int64_t *frame_starts, int64_t max_frames,
                   int64_t *n_emitted);

#ifdef __cplusplus
}
#endif
#endif /* AA_STFT_H */

/*
 * stft_oracle.c — CPU restatement of audio-analyzer-rs's STFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / the timed CPU leg.
 *
 * PARITY UNPINNED: the reference holds no test, fixture or golden vector for
 * this path (no #[test] in src/dsp/fft.rs, src/audio_io/stft.rs or
 * src/analysis/onset.rs) and it is Rust, which cannot be built in this image
 * (no cargo/rustc).  The butterflies live in un-vendored crates (realfft 3.5.0,
 * rustfft 6.4.1, num-complex 0.4.6 — Cargo.lock).  This file restates the
 * published algorithm those crates implement and is pinned instead against a
 * float64 numpy.fft.rfft truth (oracle/truth.py, tests/test_oracle.py).
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference repository root).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define AAO_OK 0
#define AAO_ERR_LEN -1
#define AAO_ERR_ARG -2

/* ---------------------------------------------------------------- window -- */

/* Periodic Hann, every operation in f32.
 * Follows src/audio_io/stft.rs:641-648 (duplicate: src/analysis/onset.rs:549-556):
 *   x = (i as f32) / (n as f32);  0.5 - 0.5 * (2.0 * PI_f32 * x).cos()           */
void aao_hann_window(int n, float *w)
{
    const float pi_f32 = 3.14159265358979323846f;
    for (int i = 0; i < n; ++i) {
        float x = (float)i / (float)n;
        float arg = 2.0f * pi_f32 * x;
        w[i] = 0.5f - 0.5f * cosf(arg);
    }
}

/* --------------------------------------------------------------- framing -- */

/* Number of frames the `while available_samples >= window_size` loop emits
 * for a gap-free stream of n_samples (src/audio_io/stft.rs:273, 436-437;
 * src/analysis/onset.rs:244, 541-542): floor((L - N) / hop) + 1, 0 if L < N. */
int64_t aao_num_frames(int64_t n_samples, int n_fft, int hop)
{
    if (n_fft <= 0 || hop <= 0 || n_samples < n_fft)
        return 0;
    return (n_samples - n_fft) / hop + 1;
}

/* ------------------------------------------------------------------ plan -- */

typedef struct {
    int n;          /* real length (even) */
    int m;          /* n / 2 : complex FFT length */
    int log2m;      /* valid when pow2 */
    int pow2;
    float *tw_re;   /* W_m^k, k < m/2 (pow2 path) or m*m table index k (generic) */
    float *tw_im;
    float *sp_re;   /* split twiddles W_n^k, k <= m/2 */
    float *sp_im;
    uint32_t *rev;  /* bit reversal (pow2 path) */
    float *buf_re, *buf_im; /* scratch m */
} aao_plan;

static int ilog2_exact(int v)
{
    int l = 0;
    while ((1 << l) < v)
        ++l;
    return ((1 << l) == v) ? l : -1;
}

/* Mirrors FftProcessor::new (src/dsp/fft.rs:14-29): plan + owned scratch.
 * Twiddles are evaluated in f64 and stored as f32, the way rustfft's
 * twiddle helper does [recollection — crate source not in this image].       */
aao_plan *aao_plan_create(int n)
{
    if (n < 2 || (n & 1))
        return NULL;
    aao_plan *p = (aao_plan *)calloc(1, sizeof(aao_plan));
    p->n = n;
    p->m = n / 2;
    p->log2m = ilog2_exact(p->m);
    p->pow2 = p->log2m >= 0;
    int m = p->m;
    int ntw = p->pow2 ? (m / 2 > 0 ? m / 2 : 1) : m;
    p->tw_re = (float *)malloc(sizeof(float) * ntw);
    p->tw_im = (float *)malloc(sizeof(float) * ntw);
    for (int k = 0; k < ntw; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)m;
        p->tw_re[k] = (float)cos(a);
        p->tw_im[k] = (float)sin(a);
    }
    p->sp_re = (float *)malloc(sizeof(float) * (m / 2 + 1));
    p->sp_im = (float *)malloc(sizeof(float) * (m / 2 + 1));
    for (int k = 0; k <= m / 2; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        p->sp_re[k] = (float)cos(a);
        p->sp_im[k] = (float)sin(a);
    }
    p->buf_re = (float *)malloc(sizeof(float) * m);
    p->buf_im = (float *)malloc(sizeof(float) * m);
    if (p->pow2) {
        p->rev = (uint32_t *)malloc(sizeof(uint32_t) * m);
        for (int i = 0; i < m; ++i) {
            uint32_t r = 0;
            for (int b = 0; b < p->log2m; ++b)
                if (i & (1 << b))
                    r |= 1u << (p->log2m - 1 - b);
            p->rev[i] = r;
        }
    }
    return p;
}

void aao_plan_destroy(aao_plan *p)
{
    if (!p)
        return;
    free(p->tw_re); free(p->tw_im); free(p->sp_re); free(p->sp_im);
    free(p->rev); free(p->buf_re); free(p->buf_im);
    free(p);
}

int aao_plan_len(const aao_plan *p) { return p ? p->n : 0; }

/* ----------------------------------------------------------- complex FFT -- */

/* In-place radix-2 decimation-in-time on split re/im arrays that are already
 * in bit-reversed order.  sign = -1 forward, +1 inverse (conjugate twiddles). */
static void cfft_pow2(const aao_plan *p, float *re, float *im, int sign)
{
    const int m = p->m;
    for (int half = 1; half < m; half <<= 1) {
        const int step = m / (2 * half); /* twiddle stride */
        for (int base = 0; base < m; base += 2 * half) {
            float *ar = re + base, *ai = im + base;
            float *br = ar + half, *bi = ai + half;
            for (int j = 0; j < half; ++j) {
                float wr = p->tw_re[j * step];
                float wi = (sign < 0) ? p->tw_im[j * step] : -p->tw_im[j * step];
                float tr = br[j] * wr - bi[j] * wi;
                float ti = br[j] * wi + bi[j] * wr;
                br[j] = ar[j] - tr;
                bi[j] = ai[j] - ti;
                ar[j] = ar[j] + tr;
                ai[j] = ai[j] + ti;
            }
        }
    }
}

/* O(m^2) DFT for non-power-of-two half lengths (correctness only; the
 * reference only ever instantiates 2048 and 256 — stft.rs:169-170,
 * onset.rs:122-123).  Accumulates in f64 so it also serves as a cross-check. */
static void cdft_generic(const aao_plan *p, const float *xr, const float *xi,
                         float *yr, float *yi, int sign)
{
    const int m = p->m;
    for (int k = 0; k < m; ++k) {
        double sr = 0.0, si = 0.0;
        for (int j = 0; j < m; ++j) {
            int idx = (int)(((int64_t)j * k) % m);
            double wr = p->tw_re[idx], wi = (sign < 0) ? p->tw_im[idx] : -p->tw_im[idx];
            sr += (double)xr[j] * wr - (double)xi[j] * wi;
            si += (double)xr[j] * wi + (double)xi[j] * wr;
        }
        yr[k] = (float)sr;
        yi[k] = (float)si;
    }
}

/* ------------------------------------------------------------ real  FFT -- */

/* Unnormalised forward real DFT, n/2+1 interleaved (re,im) bins.
 * Contract: FftProcessor::process_forward (src/dsp/fft.rs:33-35) →
 * FftForward::process (:66-71) → realfft RealToComplex::process_with_scratch.
 * Algorithm (realfft's even-length path, restated): view x[n] as m = n/2
 * complex samples z[j] = x[2j] + i x[2j+1], run an m-point complex FFT, then
 *   X[k] = (Z[k] + conj(Z[m-k]))/2 - i W_n^k (Z[k] - conj(Z[m-k]))/2 ,
 * X[0] = Re Z0 + Im Z0, X[m] = Re Z0 - Im Z0 (both with zero imaginary part). */
int aao_rfft_forward(aao_plan *p, const float *in, float *out)
{
    if (!p || !in || !out)
        return AAO_ERR_ARG;
    const int m = p->m;
    float *zr = p->buf_re, *zi = p->buf_im;
    if (p->pow2) {
        for (int j = 0; j < m; ++j) {
            uint32_t r = p->rev[j];
            zr[r] = in[2 * j];
            zi[r] = in[2 * j + 1];
        }
        cfft_pow2(p, zr, zi, -1);
    } else {
        float *tr = (float *)malloc(sizeof(float) * 2 * m), *ti = tr + m;
        for (int j = 0; j < m; ++j) {
            tr[j] = in[2 * j];
            ti[j] = in[2 * j + 1];
        }
        cdft_generic(p, tr, ti, zr, zi, -1);
        free(tr);
    }
    out[0] = zr[0] + zi[0];
    out[1] = 0.0f;
    out[2 * m] = zr[0] - zi[0];
    out[2 * m + 1] = 0.0f;
    for (int k = 1; k <= m / 2; ++k) {
        int kk = m - k;
        if (kk < k)
            break;
        float ar = zr[k], ai = zi[k];
        float br = zr[kk], bi = -zi[kk]; /* conj(Z[m-k]) */
        float er = 0.5f * (ar + br), ei = 0.5f * (ai + bi);
        float orr = 0.5f * (ar - br), oi = 0.5f * (ai - bi);
        float wr = p->sp_re[k], wi = p->sp_im[k];
        /* t = -i * W * O */
        float pr = wr * orr - wi * oi, pi = wr * oi + wi * orr; /* W*O */
        float tr = pi, ti = -pr;
        out[2 * k] = er + tr;
        out[2 * k + 1] = ei + ti;
        /* X[m-k] = conj(E - t) */
        out[2 * kk] = er - tr;
        out[2 * kk + 1] = -(ei - ti);
    }
    return AAO_OK;
}

/* Unnormalised inverse (ifft(fft(x)) = n·x): FftProcessor::process_inverse
 * (src/dsp/fft.rs:39-41, 96-101).  Never called in the reference; kept for
 * boundary completeness.  Imaginary parts of X[0] and X[n/2] are ignored
 * (realfft reports an error for non-zero values there [recollection]; the
 * status return of 1 flags that case without aborting).                        */
int aao_rfft_inverse(aao_plan *p, const float *spec, float *out)
{
    if (!p || !spec || !out)
        return AAO_ERR_ARG;
    const int m = p->m;
    float *zr = p->buf_re, *zi = p->buf_im;
    int warn = (spec[1] != 0.0f || spec[2 * m + 1] != 0.0f) ? 1 : 0;
    /* Z[k] = (X[k] + conj(X[m-k])) + i conj(W_n^k) (X[k] - conj(X[m-k])) */
    float *tr = (float *)malloc(sizeof(float) * 2 * m), *ti = tr + m;
    for (int k = 0; k < m; ++k) {
        int kk = m - k;
        float ar = spec[2 * k], ai = (k == 0) ? 0.0f : spec[2 * k + 1];
        float br = spec[2 * kk], bi = (kk == m) ? 0.0f : -spec[2 * kk + 1];
        float er = ar + br, ei = ai + bi;
        float orr = ar - br, oi = ai - bi;
        double a = 2.0 * M_PI * (double)k / (double)p->n;
        float wr = (float)cos(a), wi = (float)sin(a);
        float pr = wr * orr - wi * oi, pi = wr * oi + wi * orr;
        /* i * (pr + i pi) = -pi + i pr */
        tr[k] = er - pi;
        ti[k] = ei + pr;
    }
    if (p->pow2) {
        for (int j = 0; j < m; ++j) {
            uint32_t r = p->rev[j];
            zr[r] = tr[j];
            zi[r] = ti[j];
        }
        cfft_pow2(p, zr, zi, +1);
    } else {
        cdft_generic(p, tr, ti, zr, zi, +1);
    }
    free(tr);
    for (int j = 0; j < m; ++j) {
        out[2 * j] = zr[j];
        out[2 * j + 1] = zi[j];
    }
    return warn;
}

/* ------------------------------------------------------------- magnitude -- */

/* mag[k] = c.norm() (src/audio_io/stft.rs:314-318, src/analysis/onset.rs:271-272);
 * num-complex's norm() is re.hypot(im) [recollection]; no scaling by N or by
 * the window gain.                                                             */
void aao_magnitude(int nbins, const float *spec, float *mag)
{
    for (int k = 0; k < nbins; ++k)
        mag[k] = hypotf(spec[2 * k], spec[2 * k + 1]);
}

/* ------------------------------------------------------------ batch STFT -- */

/* The frame loop of STFT::detect_pitches (src/audio_io/stft.rs:273-318) and
 * OnsetDetector::detect_onsets (src/analysis/onset.rs:244-272) for a gap-free
 * buffer: frame t = x[t*hop .. t*hop + n), multiplied by the window with a
 * single f32 rounding (stft.rs:296-299), forward FFT (:306), then either the
 * complex bins or norm() magnitudes.  Frames [f0, f1) only, rows written from
 * out[0]; returns the number of frames written or a negative error.
 * kind: 0 = complex (re,im interleaved, n/2+1 pairs per row), 1 = magnitude.   */
int64_t aao_stft_range(const float *x, int64_t n_samples, int n_fft, int hop,
                       const float *window, int kind, int64_t f0, int64_t f1,
                       float *out)
{
    if (!x || !out || !window || hop <= 0)
        return AAO_ERR_ARG;
    aao_plan *p = aao_plan_create(n_fft);
    if (!p)
        return AAO_ERR_LEN;
    int64_t nf = aao_num_frames(n_samples, n_fft, hop);
    if (f0 < 0 || f1 > nf || f0 > f1) {
        aao_plan_destroy(p);
        return AAO_ERR_ARG;
    }
    const int nb = n_fft / 2 + 1;
    float *tw = (float *)malloc(sizeof(float) * n_fft);
    float *spec = (float *)malloc(sizeof(float) * 2 * nb);
    for (int64_t t = f0; t < f1; ++t) {
        const float *src = x + t * (int64_t)hop;
        for (int i = 0; i < n_fft; ++i)
            tw[i] = src[i] * window[i];
        if (kind == 0) {
            aao_rfft_forward(p, tw, out + (t - f0) * (int64_t)(2 * nb));
        } else {
            aao_rfft_forward(p, tw, spec);
            aao_magnitude(nb, spec, out + (t - f0) * (int64_t)nb);
        }
    }
    free(tw);
    free(spec);
    aao_plan_destroy(p);
    return f1 - f0;
}

/* ------------------------------------------------- streaming ring buffer -- */

/* Restatement of the ingest + ring-buffer + hop-advance logic
 * (src/audio_io/stft.rs:198-201, 240-266, 273, 296-299, 436-437;
 * src/analysis/onset.rs:143-146, 216-237, 244, 254-257, 541-542):
 * slots are appended sample by sample with a modulo write index; if more than
 * ring_len samples are pending the read index jumps forward (oldest samples
 * dropped); frames are emitted while available >= n_fft.  What is recorded
 * for each emitted frame is the ABSOLUTE index of its first sample in the
 * input stream, which is what "bit-exact frame indexing" is tested on.        */
typedef struct {
    int n_fft, hop, ring_len;
    float *ring;
    int64_t write_pos, read_pos, available;
    int64_t abs_read; /* absolute stream index of ring[read_pos] */
} aao_stream;

aao_stream *aao_stream_create(int n_fft, int hop, int ring_len)
{
    if (n_fft <= 0 || hop <= 0 || ring_len < n_fft)
        return NULL;
    aao_stream *s = (aao_stream *)calloc(1, sizeof(aao_stream));
    s->n_fft = n_fft;
    s->hop = hop;
    s->ring_len = ring_len;
    s->ring = (float *)calloc(ring_len, sizeof(float));
    return s;
}

void aao_stream_destroy(aao_stream *s)
{
    if (s) {
        free(s->ring);
        free(s);
    }
}

/* Push one slot (stft.rs:240-266), then drain frames (stft.rs:273-437).
 * frame_starts receives the absolute first-sample index of each emitted
 * frame; frames (optional, may be NULL) receives the un-windowed n_fft samples
 * of each frame.  Returns the number of frames emitted (<= max_frames).       */
int64_t aao_stream_push(aao_stream *s, const float *slot, int slot_len,
                        int64_t *frame_starts, float *frames, int64_t max_frames)
{
    for (int i = 0; i < slot_len; ++i) {
        s->ring[s->write_pos] = slot[i];
        s->write_pos = (s->write_pos + 1) % s->ring_len;
        s->available += 1;
    }
    if (s->available > s->ring_len) {
        int64_t excess = s->available - s->ring_len;
        s->read_pos = (s->read_pos + excess) % s->ring_len;
        s->abs_read += excess;
        s->available = s->ring_len;
    }
    int64_t emitted = 0;
    while (s->available >= s->n_fft && emitted < max_frames) {
        if (frame_starts)
            frame_starts[emitted] = s->abs_read;
        if (frames)
            for (int i = 0; i < s->n_fft; ++i)
                frames[emitted * (int64_t)s->n_fft + i] =
                    s->ring[(s->read_pos + i) % s->ring_len];
        s->read_pos = (s->read_pos + s->hop) % s->ring_len;
        s->abs_read += s->hop;
        s->available -= s->hop;
        ++emitted;
    }
    return emitted;
}

/* ------------------------------------------- onset front: flux and energy -- */

/* Spectral flux and frame energy of OnsetDetector::detect_onsets
 * (src/analysis/onset.rs:261-291), sequential f32 sums in bin order:
 *   energy   += mag[i]
 *   smoothed  = i == 0 || i >= n-1 ? mag[i] : (mag[i-1] + mag[i] + mag[i+1]) / 3
 *   weight    = 1 - i / n
 *   diff      = smoothed - prev[i];  if diff > 0: flux += diff * weight
 *   prev[i]   = mag[i]                     (prev starts at zero, onset.rs:146)
 * mags: [n_frames x n_bins]; prev0: n_bins values carried in from the frame
 * before the first one (NULL = zeros).                                          */
void aao_onset_flux(const float *mags, int64_t n_frames, int n_bins, const float *prev0,
                    float *flux, float *energy)
{
    for (int64_t t = 0; t < n_frames; ++t) {
        const float *m = mags + t * (int64_t)n_bins;
        const float *pv = t == 0 ? prev0 : m - n_bins;
        float fl = 0.0f, en = 0.0f;
        for (int i = 0; i < n_bins; ++i) {
            float mag = m[i];
            en += mag;
            float sm = (i == 0 || i >= n_bins - 1) ? mag : (m[i - 1] + m[i] + m[i + 1]) / 3.0f;
            float weight = 1.0f - ((float)i / (float)n_bins);
            float diff = sm - (pv ? pv[i] : 0.0f);
            if (diff > 0.0f)
                fl += diff * weight;
        }
        flux[t] = fl;
        energy[t] = en;
    }
}

/* Test support, no reference counterpart: the product's flux kernel (csrc/aa_mel.hip, flux_div3) forms the
 * reference's `/ 3.0` (onset.rs:268) as q0 = x * RN(1/3); e = fma(-3, q0, x); q = fma(e, RN(1/3), q0), followed by the
 * hardware's division fix-up for the inputs the remainder step cannot serve (+-inf, whose remainder is NaN, and -0, whose
 * sign the final add loses).  Counts the f32 bit patterns in [first, first + count) for which the three-operation
 * core differs from the division (NaNs compare equal) and records the first `cap` of them: the whole 2^32 sweep must
 * come back with exactly those three.                                                                          */
__attribute__((target("fma"))) uint64_t aao_div3_mismatches(uint32_t first, uint64_t count, uint32_t *out, int cap)
{
    const float r = 0x1.555556p-2f;
    uint64_t bad = 0;
    for (uint64_t i = 0; i < count; ++i) {
        uint32_t bits = first + (uint32_t)i, ba, bb;
        float x;
        memcpy(&x, &bits, sizeof x);
        const float q0 = x * r;
        const float e = __builtin_fmaf(-3.0f, q0, x);
        const float a = __builtin_fmaf(e, r, q0), b = x / 3.0f;
        memcpy(&ba, &a, sizeof ba);
        memcpy(&bb, &b, sizeof bb);
        if (ba != bb && !(a != a && b != b)) {
            if (out && bad < (uint64_t)cap)
                out[bad] = bits;
            ++bad;
        }
    }
    return bad;
}

/* Rest of the onset detector's per-frame logic (src/analysis/onset.rs:298-371, 531-535) after
 * the flux: per-bin "rise-once, suppress" burst floor (:311-332), silence gate (:337-339),
 * energy EMA (:345-350), FluxTracker::update (:66-83, constructed :158 with (1.5, 0.84, 0.89)),
 * combined trigger (:355-357), energy_rising (:373), velocity (:389-390) and the
 * frames_since_onset gate (:402, 531-535).  What the decision takes from the MusicalTransport are arguments:
 *   mask[t] (NULL = 0)  bit 0: the nearest metronome tick is inside the tick guard (:393-396: suppressed_by_tick);
 *                       bit 1: the calibration click has been played and an onset stamped at this frame would leave a
 *                              plausible residual, 0 .. 0.5 s behind it (:405-431: target != 0, not rejected);
 *                       bit 2: the calibration timeout has run out by this frame (:361-371)
 *   calibrated          0: calibration running, click not played yet - a gated onset is "pre-calibration noise", ignored
 *                          (:406-408): nothing fires, frames_since_onset counts on;
 *                       1: calibration_done;
 *                       2: calibrating - calibration_done is CARRIED IN THE STATE (0 for a fresh detector): a frame with
 *                          mask bit 2 sets it before its gates are evaluated (:361-371); a gated onset on a frame with
 *                          mask bit 1 sets it and is CONSUMED by the calibration - the reference sets onset_fired = true
 *                          without pushing an event (:424-428), so frames_since_onset restarts (:535-536) and the stream
 *                          goes on calibrated; a gated onset on any other frame is ignored as under 0 (target not set, or
 *                          the residual rejected :417-422).  mask == NULL under 2: every frame has bit 1, none bit 2.
 * gated = onset_detected && !suppressed_by_tick && energy_rising && frames_since_onset >= 3   (:402)
 * state: n_bins floors, then energy_ema, tracker threshold, frames_since_onset (int bits), calibration_done (int bits).
 * flags: bit0 flux_onset, bit1 bin_burst_onset, bit2 onset_detected, bit3 energy_rising,
 * bit4 onset event pushed (gated && calibration_done), bit5 suppressed_by_tick, bit6 gated,
 * bit7 gated and consumed by the calibration.  Plain f32 in the reference's operation order (-ffp-contract=off). */
void aao_onset_gates(const float *mags, int64_t n_frames, int n_bins, const float *flux,
                     const float *energy, const float *global_floor, const uint8_t *mask, int calibrated,
                     float *state, int *initialized,
                     float *out_flux, float *out_ema, float *out_excess, int32_t *out_count,
                     uint8_t *out_flags, float *out_velocity)
{
    const float ENERGY_EMA_RISE = 0.84f, ENERGY_EMA_DECAY = 0.95f;
    const float BIN_BURST_RATIO = 2.5f, FLOOR_OVERCOMPENSATE = 1.3f, FLOOR_RISE = 0.1f, FLOOR_DECAY = 0.04f;
    const float multiplier = 1.5f, rise_memory = 0.84f, decay_memory = 0.89f;
    float *fl = state;
    float ema, thr;
    int32_t since, cal_done = 0;
    if (*initialized) {
        ema = state[n_bins];
        thr = state[n_bins + 1];
        memcpy(&since, &state[n_bins + 2], sizeof since);
        memcpy(&cal_done, &state[n_bins + 3], sizeof cal_done);
    } else {
        ema = 0.0f;
        thr = 0.0f;
        since = 4;
    }
    if (calibrated != 2)
        cal_done = calibrated == 1;
    if (n_frames <= 0)
        return;                                       /* nothing seen: the state stays as it was (and uninitialised) */
    for (int64_t t = 0; t < n_frames; ++t) {
        const float *m = mags + t * (int64_t)n_bins;
        float current_flux = flux[t];
        const float frame_energy = energy[t], gf = global_floor[t];
        const float floor_eps = fmaxf(gf, 0.01f);
        if (!*initialized) {
            for (int k = 0; k < n_bins; ++k)
                fl[k] = fmaxf(m[k], gf);
            *initialized = 1;
        }
        float max_bin_excess = 0.0f;
        int32_t bin_burst_count = 0;
        for (int k = 0; k < n_bins; ++k) {
            const float mag = m[k];
            const float floor_k = fmaxf(fl[k], floor_eps);
            const float r = mag / floor_k;
            if (r > BIN_BURST_RATIO) {
                bin_burst_count += 1;
                fl[k] = mag * FLOOR_OVERCOMPENSATE;
            } else if (mag > fl[k]) {
                fl[k] += FLOOR_RISE * (mag - fl[k]);
            } else {
                fl[k] += FLOOR_DECAY * (mag - fl[k]);
            }
            if (r > max_bin_excess)
                max_bin_excess = r;
        }
        if (bin_burst_count < 2)
            current_flux = 0.0f;
        const float ema_memory = frame_energy > ema ? ENERGY_EMA_RISE : ENERGY_EMA_DECAY;
        ema = ema * ema_memory + frame_energy * (1.0f - ema_memory);
        const float memory = current_flux > thr ? rise_memory : decay_memory;
        const int is_onset = current_flux > thr;
        thr = thr * memory + current_flux * (1.0f - memory);
        if (thr < 0.9f)
            thr = 0.9f;
        const int flux_onset = is_onset && current_flux > thr * multiplier;
        const int bin_burst_onset = max_bin_excess > 3.0f && bin_burst_count >= 3;
        const int onset_detected = flux_onset && bin_burst_onset;
        const int mk = mask ? mask[t] : (calibrated == 2 ? 2 : 0);
        /* onset.rs:361-371: the timeout completes the calibration in front of the frame's decision */
        if (calibrated == 2 && (mk & 4))
            cal_done = 1;
        const int energy_rising = frame_energy > ema * 1.5f;
        /* onset.rs:393-396: suppressed_by_tick = tick_dist < tick_guard_beats (the transport's figure: an input here) */
        const int tick = onset_detected && (mk & 1);
        const int gated = onset_detected && !tick && energy_rising && since >= 3;       /* :402 */
        int fired = 0, consumed = 0;
        if (gated) {
            if (cal_done) {
                fired = 1;                            /* :433-446: the event is pushed */
            } else if (calibrated == 2 && (mk & 2)) {
                consumed = 1;                         /* :424-428: calibration_done = true; onset_fired = true */
                cal_done = 1;
            }                                         /* else :406-408 / :417-422: ignored */
        }
        float vel = fmaxf(current_flux, max_bin_excess * 5.0f) / 50.0f;
        vel = vel < 0.0f ? 0.0f : (vel > 1.0f ? 1.0f : vel);
        if (fired || consumed || (onset_detected && since < 3))                        /* :535-539 */
            since = 0;
        else if (since < INT32_MAX)
            since += 1;
        out_flux[t] = current_flux;
        out_ema[t] = ema;
        out_excess[t] = max_bin_excess;
        out_count[t] = bin_burst_count;
        out_flags[t] = (uint8_t)(flux_onset | (bin_burst_onset << 1) | (onset_detected << 2) |
                                 (energy_rising << 3) | (fired << 4) | (tick << 5) | (gated << 6) | (consumed << 7));
        out_velocity[t] = vel;
    }
    state[n_bins] = ema;
    state[n_bins + 1] = thr;
    memcpy(&state[n_bins + 2], &since, sizeof since);
    memcpy(&state[n_bins + 3], &cal_done, sizeof cal_done);
}

/* -------------------------------------- tuner front: per-bin noise floor -- */

/* Variance-aware per-bin noise floor of STFT::detect_pitches
 * (src/audio_io/stft.rs:206-224 constants, :326-367 update): a first-order recurrence over
 * frames for every bin.  state = {floor[n], prev_mag[n], volatility[n]} (3*n floats) plus the
 * `initialized` flag; global_floor = 10^(noise_floor_db/20) * half_size / 2 is supplied by the
 * caller (one value per frame, stft.rs:322-324).  eff[t][k] = min(floor[k], 2.5*global_floor).
 * Plain f32, one rounding per operation as Rust evaluates it (build with -ffp-contract=off). */
void aao_tuner_floor(const float *mags, int64_t n_frames, int n_bins, const float *global_floor,
                     float *state, int *initialized, float *eff)
{
    const float FLOOR_BASE_ALPHA = 0.04f, FLOOR_FAST_ALPHA = 0.35f, FLOOR_RELEASE = 0.02f;
    const float VOL_MEMORY = 0.75f, NOTE_RATIO = 1.5f, NOTE_VOL_MAX = 0.15f;
    float *floor_ = state, *prev = state + n_bins, *vol = state + 2 * n_bins;
    for (int64_t t = 0; t < n_frames; ++t) {
        const float *m = mags + t * (int64_t)n_bins;
        const float gf = global_floor[t];
        if (!*initialized) {
            for (int k = 0; k < n_bins; ++k) {
                floor_[k] = fmaxf(m[k], gf * 5.0f);
                prev[k] = m[k];
            }
            *initialized = 1;
        } else {
            for (int k = 0; k < n_bins; ++k) {
                float mag = m[k], fl = floor_[k];
                float delta = fabsf(mag - prev[k]);
                vol[k] = vol[k] * VOL_MEMORY + delta * (1.0f - VOL_MEMORY);
                prev[k] = mag;
                float above_ratio = mag / fmaxf(fl, 0.01f);
                float vn = vol[k] / fmaxf(mag, 0.05f);
                vn = vn < 0.0f ? 0.0f : (vn > 1.0f ? 1.0f : vn);
                int sustained = above_ratio > NOTE_RATIO && vn < NOTE_VOL_MAX;
                if (!sustained) {
                    float alpha = mag > fl ? FLOOR_BASE_ALPHA + (FLOOR_FAST_ALPHA - FLOOR_BASE_ALPHA) * vn
                                           : FLOOR_RELEASE;
                    floor_[k] = fl + alpha * (mag - fl);
                }
            }
        }
        float *e = eff + t * (int64_t)n_bins;
        for (int k = 0; k < n_bins; ++k)
            e[k] = fminf(floor_[k], gf * 2.5f);
    }
}

/* ------------------------------------ tuner front: multi-pitch extraction -- */

/* STFT::extract_pitches (src/audio_io/stft.rs:443-620) for every frame: peak picking against the
 * effective noise floor, parabolic (log-magnitude) interpolation, harmonic-sum scoring over up
 * to 14 harmonics, 50 % score cutoff, harmonic-ghost suppression, sort by score, 2-bin dedupe,
 * at most 8 notes inside [min_freq, max_freq].  f32 in the reference's operation order; ln / log2
 * are libm's logf / log2f.  One deviation, stated: the reference sorts with sort_unstable_by
 * (order of exactly equal scores unspecified); here equal scores keep ascending bin order.
 * out_freq / out_score: [n_frames][8], unused slots 0; out_count: notes per frame. */
#define AAO_MAX_NOTES 8
#define AAO_MAX_HARMONICS 14
/* rel. distance of a to the boundary b of a comparison the frame's result hangs on */
static inline void tie_margin(float *margin, float a, float b)
{
    const float d = fabsf(a - b) / fmaxf(fabsf(b), 1e-30f);
    if (d < *margin)
        *margin = d;
}
/* out_margin (nullable; test support): per frame the smallest relative distance of any float the frame's decisions
 * hang on to the boundary of its comparison - harmonic windows (expected_f against the integers and the spectrum's
 * end), the 50 % cutoff, the ghost test, the score order, the 2-bin dedupe, the frequency range.  Comparisons of INPUT
 * values alone (peak tests, 5 x / 15 x floor) are not counted: they cannot differ between two evaluations.  A frame in
 * which a 1-ulp difference of logf / log2f can change the result has a margin of ~1e-7; any other frame must come out
 * of every correct evaluation with the same notes. */
static void extract_pitches_impl(const float *mags, const float *eff_floor, int64_t n_frames, int n_bins,
                                 float bin_width, float min_freq, float max_freq, float *out_freq,
                                 float *out_score, int32_t *out_count, float *out_margin)
{
    const int half_size = n_bins;
    int min_bin = (int)ceilf(min_freq / bin_width);
    if (min_bin < 1)
        min_bin = 1;
    int max_bin = (int)floorf(max_freq / bin_width);
    if (max_bin > half_size - 2)
        max_bin = half_size - 2;
    unsigned char *is_peak = (unsigned char *)malloc((size_t)half_size);
    int *peak_bins = (int *)malloc(sizeof(int) * (size_t)half_size);
    float *scores = (float *)malloc(sizeof(float) * (size_t)half_size);
    float *frac_bins = (float *)malloc(sizeof(float) * (size_t)half_size);
    int *cand = (int *)malloc(sizeof(int) * (size_t)half_size);
    int *kept = (int *)malloc(sizeof(int) * (size_t)half_size);
    for (int64_t t = 0; t < n_frames; ++t) {
        const float *m = mags + t * (int64_t)n_bins, *nf = eff_floor + t * (int64_t)n_bins;
        float *of = out_freq + t * AAO_MAX_NOTES, *os = out_score + t * AAO_MAX_NOTES;
        for (int i = 0; i < AAO_MAX_NOTES; ++i)
            of[i] = os[i] = 0.0f;
        out_count[t] = 0;
        float margin = INFINITY;
        if (out_margin)
            out_margin[t] = margin;
        if (min_bin >= max_bin)
            continue;
        memset(is_peak, 0, (size_t)half_size);
        int n_peaks = 0;
        for (int k = min_bin + 1; k < max_bin; ++k)
            if (m[k] > nf[k] && m[k] >= m[k - 1] && m[k] >= m[k + 1]) {
                is_peak[k] = 1;
                peak_bins[n_peaks++] = k;
            }
        if (n_peaks == 0)
            continue;
        for (int p = 0; p < n_peaks; ++p) {
            const int k = peak_bins[p];
            const float fund_mag = m[k];
            scores[k] = 0.0f;
            frac_bins[k] = 0.0f;
            if (fund_mag < nf[k] * 5.0f)
                continue;
            float frac_bin = (float)k;
            if (k >= 1 && k + 1 < half_size) {
                const float y_l = logf(m[k - 1]), y_c = logf(m[k]), y_r = logf(m[k + 1]);
                const float denom = y_l - 2.0f * y_c + y_r;
                float delta = 0.0f;
                if (!(fabsf(denom) < 1e-30f)) {
                    delta = 0.5f * (y_l - y_r) / denom;
                    delta = delta < -1.0f ? -1.0f : (delta > 1.0f ? 1.0f : delta);
                } else {
                    margin = 0.0f;                      /* (a vanishing curvature: the branch itself is the tie) */
                }
                frac_bin = (float)k + delta;
            }
            frac_bins[k] = frac_bin;
            float score = fund_mag;
            int last = k, longest_run = 0, current_run = 0, total_harms = 0;
            for (int n = 2; n <= AAO_MAX_HARMONICS; ++n) {
                const float expected_f = frac_bin * (float)n;
                tie_margin(&margin, expected_f, (float)half_size);
                if (expected_f >= (float)half_size)
                    break;
                tie_margin(&margin, expected_f, roundf(expected_f));
                float lo = floorf(expected_f - 1.0f), hi = ceilf(expected_f + 1.0f);
                int search_start = lo > 0.0f ? (int)lo : 0;
                if (search_start < last + 1)
                    search_start = last + 1;
                int search_end = (int)hi;
                if (search_end > half_size - 1)
                    search_end = half_size - 1;
                int best_hbin = 0;
                float best_mag = 0.0f;
                for (int h = search_start; h <= search_end; ++h)
                    if (is_peak[h] && m[h] > best_mag) {
                        best_mag = m[h];
                        best_hbin = h;
                    }
                if (best_hbin != 0) {
                    score += best_mag;
                    last = best_hbin;
                    current_run += 1;
                    total_harms += 1;
                } else {
                    if (current_run > longest_run)
                        longest_run = current_run;
                    current_run = 0;
                }
            }
            if (current_run > longest_run)
                longest_run = current_run;
            if (longest_run < 3 && fund_mag < 15.0f * nf[k]) {
                scores[k] = 0.0f;
            } else {
                const float log_score = log2f(0.5f + score);
                const float struct_mult =
                    (1.0f + (float)longest_run + (float)total_harms / 2.0f) / (1.0f + (float)AAO_MAX_HARMONICS);
                scores[k] = log_score * struct_mult;
            }
        }
        float max_score = 0.0f;
        for (int p = 0; p < n_peaks; ++p)
            max_score = fmaxf(max_score, scores[peak_bins[p]]);
        if (max_score == 0.0f)
            continue;
        const float cutoff = max_score * 0.5f;
        int n_cand = 0;
        for (int p = 0; p < n_peaks; ++p) {
            if (scores[peak_bins[p]] != 0.0f)
                tie_margin(&margin, scores[peak_bins[p]], cutoff);
            if (scores[peak_bins[p]] >= cutoff)
                cand[n_cand++] = peak_bins[p];
        }
        /* harmonic ghosts: freq_i ~ N freq_j (N = 2..5) with a score below 1.05 x that of j */
        int n_kept = 0;
        for (int i = 0; i < n_cand; ++i) {
            const float freq_i = frac_bins[cand[i]] * bin_width, score_i = scores[cand[i]];
            int suppressed = 0;
            for (int j = 0; j < n_cand && !suppressed; ++j) {
                if (i == j)
                    continue;
                const float freq_j = frac_bins[cand[j]] * bin_width, score_j = scores[cand[j]];
                const float ratio = freq_i / freq_j;
                const float nearest = roundf(ratio);
                if (out_margin && ratio > 1.4f && ratio < 5.6f) {
                    tie_margin(&margin, ratio, floorf(ratio) + 0.5f);
                    if (nearest >= 2.0f && nearest <= 5.0f) {
                        tie_margin(&margin, fabsf(ratio / nearest - 1.0f), 0.03f);
                        tie_margin(&margin, score_i, score_j * 1.05f);
                    }
                }
                if (nearest >= 2.0f && nearest <= 5.0f && fabsf(ratio / nearest - 1.0f) < 0.03f &&
                    score_i < score_j * 1.05f)
                    suppressed = 1;
            }
            if (!suppressed)
                kept[n_kept++] = cand[i];
        }
        for (int i = 0; out_margin && i < n_kept; ++i)
            for (int j = i + 1; j < n_kept; ++j)
                if (scores[kept[i]] != scores[kept[j]])
                    tie_margin(&margin, scores[kept[i]], scores[kept[j]]);
        /* descending score, stable (insertion sort) */
        for (int i = 1; i < n_kept; ++i) {
            const int b = kept[i];
            int j = i - 1;
            while (j >= 0 && scores[kept[j]] < scores[b]) {
                kept[j + 1] = kept[j];
                --j;
            }
            kept[j + 1] = b;
        }
        int acc[AAO_MAX_NOTES], n_acc = 0, n_dedup = 0;
        for (int i = 0; i < n_kept && n_dedup < AAO_MAX_NOTES; ++i) {
            const float frac_i = frac_bins[kept[i]];
            int conflict = 0;
            for (int j = 0; j < n_acc; ++j) {
                tie_margin(&margin, fabsf(frac_i - frac_bins[acc[j]]), 2.0f);
                if (fabsf(frac_i - frac_bins[acc[j]]) < 2.0f)
                    conflict = 1;
            }
            if (!conflict) {
                acc[n_acc++] = kept[i];
                ++n_dedup;
            }
        }
        int n_out = 0;
        for (int i = 0; i < n_acc; ++i) {
            const float freq = frac_bins[acc[i]] * bin_width;
            tie_margin(&margin, freq, min_freq);
            tie_margin(&margin, freq, max_freq);
            if (freq >= min_freq && freq <= max_freq) {
                of[n_out] = freq;
                os[n_out] = scores[acc[i]];
                ++n_out;
            }
        }
        out_count[t] = n_out;
        if (out_margin)
            out_margin[t] = margin;
    }
    free(is_peak);
    free(peak_bins);
    free(scores);
    free(frac_bins);
    free(cand);
    free(kept);
}
void aao_extract_pitches(const float *mags, const float *eff_floor, int64_t n_frames, int n_bins,
                         float bin_width, float min_freq, float max_freq, float *out_freq,
                         float *out_score, int32_t *out_count)
{
    extract_pitches_impl(mags, eff_floor, n_frames, n_bins, bin_width, min_freq, max_freq, out_freq, out_score,
                         out_count, NULL);
}
void aao_extract_pitches_margin(const float *mags, const float *eff_floor, int64_t n_frames, int n_bins,
                                float bin_width, float min_freq, float max_freq, float *out_freq,
                                float *out_score, int32_t *out_count, float *out_margin)
{
    extract_pitches_impl(mags, eff_floor, n_frames, n_bins, bin_width, min_freq, max_freq, out_freq, out_score,
                         out_count, out_margin);
}

/* PitchTracker::process (src/audio_io/stft.rs:28-117) over a sequence of frames: hit / miss
 * hysteresis on note tracks (display_threshold 2, max_life 3, tolerance 3 %), EMA-blended or
 * (on an onset) snapped frequencies.  tracks: [cap][3] floats {freq, score, life}, *n_tracks in /
 * out; out_*: [n_frames][cap] stable pitches per frame. */
void aao_pitch_tracker(const float *freq, const float *score, const int32_t *count, const uint8_t *onset,
                       int64_t n_frames, float *tracks, int *n_tracks, int cap, float *out_freq,
                       float *out_score, int32_t *out_count)
{
    const int display_threshold = 2, max_life = 3;
    const float tolerance = 0.03f;
    float *tf = (float *)malloc(sizeof(float) * (size_t)(cap + 8)), *ts = (float *)malloc(sizeof(float) * (size_t)(cap + 8));
    int *life = (int *)malloc(sizeof(int) * (size_t)(cap + 8));
    unsigned char *matched = (unsigned char *)malloc((size_t)(cap + 8));
    int n = *n_tracks;
    for (int i = 0; i < n; ++i) {
        tf[i] = tracks[3 * i];
        ts[i] = tracks[3 * i + 1];
        life[i] = (int)tracks[3 * i + 2];
    }
    for (int64_t t = 0; t < n_frames; ++t) {
        const int is_onset = onset ? onset[t] != 0 : 0;
        memset(matched, 0, (size_t)(cap + 8));
        for (int r = 0; r < count[t] && r < 8; ++r) {
            const float raw_freq = freq[t * 8 + r], raw_score = score[t * 8 + r];
            int found = 0;
            for (int i = 0; i < n; ++i) {
                if (matched[i])
                    continue;
                if (fabsf(tf[i] - raw_freq) / tf[i] < tolerance) {
                    tf[i] = is_onset ? raw_freq : tf[i] * 0.6f + raw_freq * 0.4f;
                    ts[i] = raw_score;
                    life[i] = life[i] + 1 < max_life ? life[i] + 1 : max_life;
                    matched[i] = 1;
                    found = 1;
                    break;
                }
            }
            if (!found && n < cap + 8) {
                tf[n] = raw_freq;
                ts[n] = raw_score;
                life[n] = 1;
                matched[n] = 1;
                ++n;
            }
        }
        int n_out = 0, w = 0;
        for (int i = 0; i < n; ++i) {
            if (!matched[i])
                life[i] = is_onset ? 0 : life[i] - 1;
            if (life[i] <= 0)
                continue;                     /* destroyed */
            if (life[i] >= display_threshold && n_out < cap) {
                out_freq[t * cap + n_out] = tf[i];
                out_score[t * cap + n_out] = ts[i];
                ++n_out;
            }
            tf[w] = tf[i];
            ts[w] = ts[i];
            life[w] = life[i];
            ++w;
        }
        n = w;
        for (int i = n_out; i < cap; ++i)
            out_freq[t * cap + i] = out_score[t * cap + i] = 0.0f;
        out_count[t] = n_out;
    }
    if (n > cap)
        n = cap;
    for (int i = 0; i < n; ++i) {
        tracks[3 * i] = tf[i];
        tracks[3 * i + 1] = ts[i];
        tracks[3 * i + 2] = (float)life[i];
    }
    *n_tracks = n;
    free(tf);
    free(ts);
    free(life);
    free(matched);
}

/* ---------------------------------- input conditioning (reducer thread) -- */

/* The step before the path ("next" row f-3): what the reducer thread does to every 1024-sample slot
 * before the STFT workers see it (src/audio_io/mod.rs:351-478): 40 Hz high-pass and 14 kHz low-pass
 * biquads (Q 0.707), envelope gate (-60 dB threshold, 40 ms release, 20 ms hold, ratio^4), then
 * DynamicsTracker::process_slot (src/audio_io/dynamics.rs:152-360; constructed mod.rs:341-349 with
 * target -18 dB, max boost 100 dB, smoothing 240 s): slot RMS, noise floor = p10 of the quiet
 * slots' RMS (256 history), active / broadband (kurtosis) classification, p50 / p95 of the
 * playing slots' RMS (5000 history), smoothed AGC gain with a 0.97 peak clamp, dynamic level.
 * Plain f32 in the reference's order; libm for cos / sin / exp / pow / log10 / sqrt. */
typedef struct {
    float hp[5], lp[5];                 /* b0 b1 b2 a1 a2 (normalised by a0) */
    float hp_x1, hp_x2, hp_y1, hp_y2, lp_x1, lp_x2, lp_y1, lp_y2;
    float gate_threshold_linear, envelope, release_coeff;
    int64_t gate_hold_samples, gate_hold_remaining;
    int slot_len;
    float long_history[256], play_history[5000], sort_buf[5000];
    int long_pos, long_filled, play_pos, play_filled;
    float current_gain_linear, target_db, max_boost_db, smooth_alpha, silence_decay_alpha;
    float active_snr_db, bootstrap_floor_db;
} aao_conditioner;

static void aao_biquad(float sample_rate, float freq, int is_lpf, float *c)
{
    const float w0 = 2.0f * 3.14159265358979323846f * freq / sample_rate;
    const float cos_w0 = cosf(w0), sin_w0 = sinf(w0);
    const float alpha = sin_w0 / (2.0f * 0.707f);
    float b0, b1, b2;
    if (is_lpf) {
        b0 = (1.0f - cos_w0) / 2.0f;
        b1 = 1.0f - cos_w0;
        b2 = (1.0f - cos_w0) / 2.0f;
    } else {
        b0 = (1.0f + cos_w0) / 2.0f;
        b1 = -(1.0f + cos_w0);
        b2 = (1.0f + cos_w0) / 2.0f;
    }
    const float a0 = 1.0f + alpha, a1 = -2.0f * cos_w0, a2 = 1.0f - alpha;
    c[0] = b0 / a0;
    c[1] = b1 / a0;
    c[2] = b2 / a0;
    c[3] = a1 / a0;
    c[4] = a2 / a0;
}
static float aao_linear_to_db(float l) { return 20.0f * log10f(fmaxf(l, 1e-9f)); }
static int aao_cmp_f32(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

void *aao_conditioner_create(float sample_rate, int slot_len)
{
    aao_conditioner *c = (aao_conditioner *)calloc(1, sizeof *c);
    if (!c)
        return NULL;
    aao_biquad(sample_rate, 40.0f, 0, c->hp);
    aao_biquad(sample_rate, 14000.0f, 1, c->lp);
    c->gate_threshold_linear = powf(10.0f, -60.0f / 20.0f);
    c->release_coeff = expf(-1.0f / (0.040f * sample_rate));
    c->gate_hold_samples = (int64_t)(0.020f * sample_rate);
    c->slot_len = slot_len;
    const float slot_rate = sample_rate / (float)slot_len;
    c->current_gain_linear = 1.0f;
    c->target_db = -18.0f;
    c->max_boost_db = 100.0f;
    c->smooth_alpha = 1.0f - expf(-1.0f / (240.0f * slot_rate));
    c->silence_decay_alpha = 1.0f - expf(-1.0f / (10.0f * slot_rate));
    c->active_snr_db = 20.0f;
    c->bootstrap_floor_db = -55.0f;
    return c;
}
void aao_conditioner_destroy(void *p) { free(p); }

/* n_slots slots of slot_len samples, in place; per-slot outputs as DynamicsOutput (dynamics.rs:340-347):
 * level 0 = Silence, 1..8 = Ppp..Fff. */
void aao_conditioner_process(void *p, float *samples, int64_t n_slots, float *noise_floor_db_out,
                             float *rms_db_out, float *gain_db_out, float *median_db_out, int32_t *level_out)
{
    aao_conditioner *c = (aao_conditioner *)p;
    const int n = c->slot_len;
    for (int64_t s = 0; s < n_slots; ++s) {
        float *slot = samples + s * (int64_t)n;
        for (int i = 0; i < n; ++i) {
            float x = slot[i];
            const float hp_out = c->hp[0] * x + c->hp[1] * c->hp_x1 + c->hp[2] * c->hp_x2 - c->hp[3] * c->hp_y1 -
                                 c->hp[4] * c->hp_y2;
            c->hp_x2 = c->hp_x1;
            c->hp_x1 = x;
            c->hp_y2 = c->hp_y1;
            c->hp_y1 = hp_out;
            x = hp_out;
            const float lp_out = c->lp[0] * x + c->lp[1] * c->lp_x1 + c->lp[2] * c->lp_x2 - c->lp[3] * c->lp_y1 -
                                 c->lp[4] * c->lp_y2;
            c->lp_x2 = c->lp_x1;
            c->lp_x1 = x;
            c->lp_y2 = c->lp_y1;
            c->lp_y1 = lp_out;
            x = lp_out;
            const float abs_in = fabsf(x);
            if (abs_in > c->envelope) {
                c->envelope = abs_in;
                c->gate_hold_remaining = c->gate_hold_samples;
            } else {
                c->envelope = c->release_coeff * c->envelope + (1.0f - c->release_coeff) * abs_in;
            }
            float gain;
            if (c->envelope >= c->gate_threshold_linear) {
                gain = 1.0f;
            } else if (c->gate_hold_remaining > 0) {
                c->gate_hold_remaining -= 1;
                gain = 1.0f;
            } else {
                const float ratio = c->envelope / c->gate_threshold_linear;
                gain = ratio * ratio * ratio * ratio;
            }
            slot[i] = x * gain;
        }
        /* ---- DynamicsTracker::process_slot ---- */
        float sum_sq = 0.0f;
        for (int i = 0; i < n; ++i)
            sum_sq += slot[i] * slot[i];
        const float rms_linear = sqrtf(sum_sq / (float)n);
        const float rms_db = aao_linear_to_db(rms_linear);
        const int long_n = c->long_filled ? 256 : (c->long_pos > 1 ? c->long_pos : 1);
        memcpy(c->sort_buf, c->long_history, sizeof(float) * (size_t)long_n);
        qsort(c->sort_buf, (size_t)long_n, sizeof(float), aao_cmp_f32);
        const int p10_idx = (int)((float)(long_n - 1) * 0.10f);
        const float noise_floor_db = aao_linear_to_db(fmaxf(c->sort_buf[p10_idx], 1e-9f));
        const float floor_db = long_n >= 32 ? noise_floor_db : c->bootstrap_floor_db;
        const int is_active = rms_db > floor_db + c->active_snr_db;
        int is_broadband = 0;
        if (is_active) {
            const float mean_sq = rms_linear * rms_linear;
            float quad = 0.0f;
            for (int i = 0; i < n; ++i) {
                const float s2 = slot[i] * slot[i];
                quad += s2 * s2;
            }
            const float mean_quad = quad / (float)n;
            const float kurtosis = mean_sq > 1e-18f ? mean_quad / (mean_sq * mean_sq) : 3.0f;
            is_broadband = kurtosis >= 2.75f && kurtosis <= 3.8f && rms_db < -45.0f;
        }
        const int is_playing = is_active && !is_broadband;
        if (!is_active || is_broadband) {
            c->long_history[c->long_pos] = rms_linear;
            c->long_pos = (c->long_pos + 1) % 256;
            if (c->long_pos == 0)
                c->long_filled = 1;
        }
        if (is_playing) {
            c->play_history[c->play_pos] = rms_linear;
            c->play_pos = (c->play_pos + 1) % 5000;
            if (c->play_pos == 0)
                c->play_filled = 1;
        }
        const int play_n = c->play_filled ? 5000 : c->play_pos;
        float raw_gain_db = 0.0f, session_median_db = rms_db;
        if (play_n > 0) {
            memcpy(c->sort_buf, c->play_history, sizeof(float) * (size_t)play_n);
            qsort(c->sort_buf, (size_t)play_n, sizeof(float), aao_cmp_f32);
            const int p50_idx = (play_n - 1) / 2;
            const int p95_idx = (int)((float)(play_n - 1) * 0.95f);
            session_median_db = aao_linear_to_db(fmaxf(c->sort_buf[p50_idx], 1e-9f));
            const float p95_db = aao_linear_to_db(fmaxf(c->sort_buf[p95_idx], 1e-9f));
            float g = c->target_db - p95_db;
            raw_gain_db = g < 0.0f ? 0.0f : (g > c->max_boost_db ? c->max_boost_db : g);
        }
        if (is_playing) {
            const float target_linear = powf(10.0f, raw_gain_db / 20.0f);
            c->current_gain_linear += c->smooth_alpha * (target_linear - c->current_gain_linear);
        } else {
            c->current_gain_linear += c->silence_decay_alpha * (1.0f - c->current_gain_linear);
        }
        float peak = 0.0f;
        for (int i = 0; i < n; ++i)
            peak = fmaxf(peak, fabsf(slot[i]));
        peak = fmaxf(peak, 1e-9f);
        const float headroom_limit = 0.97f / peak;
        const float effective_gain = fminf(c->current_gain_linear, headroom_limit);
        for (int i = 0; i < n; ++i)
            slot[i] *= effective_gain;
        int32_t level = 0;
        if (is_playing) {
            const float r = rms_db - session_median_db;
            level = r < -15.0f ? 1 : r < -9.0f ? 2 : r < -4.5f ? 3 : r < -1.5f ? 4 : r < 1.5f ? 5 : r < 4.5f ? 6
                                                                                                 : r < 9.0f ? 7 : 8;
        }
        noise_floor_db_out[s] = noise_floor_db;
        rms_db_out[s] = rms_db;
        gain_db_out[s] = aao_linear_to_db(effective_gain);
        median_db_out[s] = session_median_db;
        level_out[s] = level;
    }
}

/* ------------------------------------------------------ synthetic signals -- */

/* LCG of the reference's metronome noise tick (src/generators/metronome.rs:56-57):
 *   seed = (seed.wrapping_mul(1103515245).wrapping_add(12345)) & 0x7FFFFFFF
 * (the u32 wrap is irrelevant under the 31-bit mask).                          */
uint32_t aao_lcg_next(uint32_t seed)
{
    return (uint32_t)(((uint64_t)seed * 1103515245ull + 12345ull) & 0x7FFFFFFFull);
}

/*
 * theory_oracle.c — CPU restatement of the note / interval naming behind the reference's
 * TunerOutput (src/analysis/theory.rs, src/analysis/tuner.rs:141-207).
 *
 * TEST INFRASTRUCTURE ONLY (see stft_oracle.c): only tests/ may load it, as the checker.
 *
 * PINNED: unlike the STFT path this stage has known-answer tests in the reference itself —
 * src/analysis/theory.rs:404-627 (A4 / C4 / C#4 naming, cents range, Note::new -> to_freq
 * round trips, try_new errors, five equal-temperament intervals, MidiNote round trips).
 * tests/test_theory.py restates every one of them against this file and against the product.
 *
 * Rust semantics restated here: `f32 % f32` is fmodf, `f32::round` is roundf (half away from
 * zero), `as u8` / `as usize` saturate (NaN -> 0), `u8 + 12` wraps in a release build,
 * `total_cmp` is the IEEE total order, `min_by` keeps the first of equal minima and `max_by`
 * the last of equal maxima.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define AAO_NOTES_MAX 32

static uint8_t sat_u8(float v)
{
    if (!(v > 0.0f))
        return 0; /* NaN, negatives, zero */
    if (v >= 255.0f)
        return 255;
    return (uint8_t)v;
}

static int32_t total_key(float v)
{
    int32_t b;
    memcpy(&b, &v, 4);
    return b ^ (int32_t)((uint32_t)(b >> 31) >> 1);
}

/* log-frequency position in cents above C0 (theory.rs:200-202, same lines at :17-19) */
static float cents_above_c0(float freq, float base_freq)
{
    const float base = base_freq * powf(2.0f, -4.75f);
    return log2f(freq / base) * 1200.0f;
}

static float fold_cents(float log_cents)
{
    float cents = fmodf(log_cents, 100.0f);
    return cents < 50.0f ? cents : -(100.0f - cents);
}

/* Note::from_freq + get_name + get_cents (theory.rs:199-232, 238-254).  name: >= 8 bytes. */
void aao_note_from_freq(float freq, float base_freq, char *name, float *cents)
{
    static const char *const kNames[12] = {"C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"};
    const float lg = cents_above_c0(freq, base_freq);
    const uint8_t octave = sat_u8((lg + 50.0f) / 1200.0f);
    const float s = fmodf(roundf(lg / 100.0f), 12.0f);
    const int semis = s > 0.0f ? (int)s : 0; /* as usize: NaN and negatives -> 0; |s| < 12 */
    *cents = fold_cents(lg);
    snprintf(name, 8, "%s%u", kNames[semis], (unsigned)octave);
}

/* Note::try_new + to_freq (theory.rs:104-197).  Returns 0, or -1 when the name does not parse. */
int aao_note_to_freq(const char *name, float cents, float base_freq, float *freq)
{
    const size_t len = strlen(name);
    if (len < 2)
        return -1;
    int semis;
    switch (name[0]) {
    case 'C': semis = -9; break;
    case 'D': semis = -7; break;
    case 'E': semis = -5; break;
    case 'F': semis = -4; break;
    case 'G': semis = -2; break;
    case 'A': semis = 0; break;
    case 'B': semis = 2; break;
    default: return -1;
    }
    size_t start = 2;
    if (name[1] == '#')
        semis += 1;
    else if (name[1] == 'b')
        semis -= 1;
    else if (len > 2 && name[1] == 'x')
        semis += 2;
    else if (len > 2 && name[1] == 'B')
        semis -= 2;
    else if (name[1] == 'n')
        semis += 0;
    else
        start = 1;
    /* u8::from_str: optional '+', then digits only, value <= 255 */
    const char *p = name + start;
    if (*p == '+')
        ++p;
    if (!*p)
        return -1;
    unsigned octave = 0;
    for (; *p; ++p) {
        if (*p < '0' || *p > '9')
            return -1;
        octave = octave * 10 + (unsigned)(*p - '0');
        if (octave > 255)
            return -1;
    }
    semis += ((int)octave - 4) * 12;
    *freq = base_freq * powf(2.0f, ((float)semis + cents / 100.0f) / 12.0f);
    return 0;
}

/* MidiNote::from_freq (theory.rs:16-28) and to_freq (:40-43) */
void aao_midi_from_freq(float freq, float base_freq, int32_t *midi, float *cents)
{
    const float lg = cents_above_c0(freq, base_freq);
    *cents = fold_cents(lg);
    *midi = (uint8_t)(sat_u8(roundf(lg / 100.0f)) + 12u);
}

float aao_midi_to_freq(int32_t midi, float cents, float base_freq)
{
    return base_freq * powf(2.0f, ((float)(uint8_t)midi - 69.0f + cents / 100.0f) / 12.0f);
}

/* Interval::new + get_name + get_accuracy (theory.rs:289-381).  system: 0 equal temperament,
 * 1 just intonation, 2 Pythagorean (enum order, tuner.rs:12-18).  n < 2 or f[0] == 0 gives the
 * Per8 / 0.0 default.  A non-finite ratio would spin the reference's halving loop forever; the
 * caller (aao_tuner_output) never passes one. */
void aao_interval(const float *f, int n, int system, char *name, float *accuracy)
{
    static const char *const kNames[13] = {"Per8", "Min2", "Maj2", "Min3", "Maj3", "Per4", "Aug4",
                                           "Per5", "Min6", "Maj6", "Min7", "Maj7", "Per8"};
    static const float kJust[13] = {1.0f,         16.0f / 15.0f, 9.0f / 8.0f, 6.0f / 5.0f, 5.0f / 4.0f,
                                    4.0f / 3.0f,  45.0f / 32.0f, 3.0f / 2.0f, 8.0f / 5.0f, 5.0f / 3.0f,
                                    9.0f / 5.0f,  15.0f / 8.0f,  2.0f};
    static const float kPyth[13] = {1.0f,          256.0f / 243.0f, 9.0f / 8.0f,   32.0f / 27.0f, 81.0f / 64.0f,
                                    4.0f / 3.0f,   729.0f / 512.0f, 3.0f / 2.0f,   128.0f / 81.0f, 27.0f / 16.0f,
                                    32.0f / 9.0f,  243.0f / 128.0f, 2.0f}; /* 32/9 as the reference has it */
    static const float kEqual[13] = {1.0f,    1.0595f, 1.1225f, 1.1892f, 1.2599f, 1.3348f, 1.4142f,
                                     1.4983f, 1.5874f, 1.6818f, 1.7818f, 1.8877f, 2.0f};
    if (n < 2 || f[0] == 0.0f) {
        strcpy(name, "Per8");
        *accuracy = 0.0f;
        return;
    }
    float ratio = f[1] / f[0];
    while (ratio > 2.0f && isfinite(ratio))
        ratio /= 2.0f;
    const float *r = system == 1 ? kJust : (system == 2 ? kPyth : kEqual);
    int best = 0;
    for (int i = 1; i < 13; ++i)
        if (total_key(fabsf(ratio - r[i])) < total_key(fabsf(ratio - r[best])))
            best = i;
    strcpy(name, kNames[best]);
    *accuracy = -logf(r[best] / ratio) * 1732.5f;
}

/* One iteration of Tuner::run's body (tuner.rs:153-193) for a non-empty list of n <= 32
 * (freq, score) pairs.  mode: 0 MultiPitch, 1 SinglePitch.  names: n_notes rows of 8 bytes. */
void aao_tuner_output(const float *freq, const float *score, int n, int mode, int system, float base_freq,
                      char *label, float *cents, int32_t *n_notes, char *names, float *accuracies)
{
    *cents = 0.0f;
    label[0] = 0;
    if (n > AAO_NOTES_MAX)
        n = AAO_NOTES_MAX;
    if (n == 1 || mode == 1) {
        int best = 0;
        for (int i = 1; i < n; ++i)
            if (total_key(score[i]) >= total_key(score[best]))
                best = i;
        aao_note_from_freq(freq[best], base_freq, names, accuracies);
        strcpy(label, names);
        *cents = accuracies[0];
        *n_notes = 1;
    } else if (n == 2) {
        float f[2] = {freq[0], freq[1]};
        if (total_key(f[1]) < total_key(f[0])) {
            f[0] = freq[1];
            f[1] = freq[0];
        }
        aao_interval(f, 2, system, label, cents);
        for (int i = 0; i < 2; ++i)
            aao_note_from_freq(f[i], base_freq, names + 8 * i, accuracies + i);
        *n_notes = 2;
    } else {
        for (int i = 0; i < n; ++i) {
            aao_note_from_freq(freq[i], base_freq, names + 8 * i, accuracies + i);
            if (i)
                strcat(label, " ");
            strcat(label, names + 8 * i);
        }
        *n_notes = n;
    }
}

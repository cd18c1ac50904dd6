// TunerOutput stage of the C ABI (include/aa_stft.h): what the reference's tuner thread does with
// the stable (freq, score) lists the STFT worker sends it - note and interval naming
// (src/analysis/theory.rs) and the JSON string Tuner::poll_output returns (src/lib.rs:80-82).
// A handful of scalar f32 operations per analysis frame: host code, as in the reference.
//
// Rust semantics kept: `%` on f32 is fmodf, f32::round is roundf, float -> integer `as` casts
// saturate (NaN -> 0), `u8 + 12` wraps (release build), total_cmp is the IEEE total order,
// min_by keeps the first and max_by the last of equal extrema.
#include "../../include/aa_stft.h"
#include "aa_internal.hpp"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#pragma clang fp contract(off)

namespace {

struct Pitch {
    const char *letters;   // "C", "C#", ...
};
constexpr Pitch kChromatic[12] = {{"C"}, {"C#"}, {"D"}, {"D#"}, {"E"}, {"F"}, {"F#"}, {"G"}, {"G#"}, {"A"}, {"A#"}, {"B"}};

inline int32_t total_order_key(float v)
{
    int32_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    return bits ^ (int32_t)((uint32_t)(bits >> 31) >> 1);
}

inline unsigned saturate_u8(float v)
{
    if (std::isnan(v) || v <= 0.0f)
        return 0u;
    return v >= 255.0f ? 255u : (unsigned)v;
}

// position of freq in cents above C0 for a given A4 (theory.rs:17-19, 200-202)
inline float cents_from_c0(float freq, float a4)
{
    const float c0 = a4 * std::pow(2.0f, -4.75f);
    return std::log2(freq / c0) * 1200.0f;
}

inline float nearest_semitone_offset(float cents_c0)
{
    const float r = std::fmod(cents_c0, 100.0f);
    return r < 50.0f ? r : -(100.0f - r);
}

struct NoteId {
    int chroma;
    unsigned octave;
    float cents;
};

NoteId identify(float freq, float a4)
{
    const float pos = cents_from_c0(freq, a4);
    NoteId id;
    id.octave = saturate_u8((pos + 50.0f) / 1200.0f);
    const float step = std::fmod(std::round(pos / 100.0f), 12.0f);
    id.chroma = (std::isnan(step) || step <= 0.0f) ? 0 : (int)step;
    id.cents = nearest_semitone_offset(pos);
    return id;
}

void write_name(const NoteId &id, char *name)
{
    std::snprintf(name, AA_NOTE_NAME_LEN, "%s%u", kChromatic[id.chroma].letters, id.octave);
}

struct IntervalId {
    const char *name;
    float accuracy;
};

IntervalId classify_interval(const float *f, int n, int system)
{
    static const char *const names[13] = {"Per8", "Min2", "Maj2", "Min3", "Maj3", "Per4", "Aug4",
                                          "Per5", "Min6", "Maj6", "Min7", "Maj7", "Per8"};
    // theory.rs:303-340; the Pythagorean minor seventh is 32/9 there (sic), kept: it can never be nearest
    static const float just[13] = {1.0f, 16.0f / 15.0f, 9.0f / 8.0f, 6.0f / 5.0f, 5.0f / 4.0f, 4.0f / 3.0f,
                                   45.0f / 32.0f, 3.0f / 2.0f, 8.0f / 5.0f, 5.0f / 3.0f, 9.0f / 5.0f, 15.0f / 8.0f, 2.0f};
    static const float pyth[13] = {1.0f, 256.0f / 243.0f, 9.0f / 8.0f, 32.0f / 27.0f, 81.0f / 64.0f, 4.0f / 3.0f,
                                   729.0f / 512.0f, 3.0f / 2.0f, 128.0f / 81.0f, 27.0f / 16.0f, 32.0f / 9.0f,
                                   243.0f / 128.0f, 2.0f};
    static const float equal[13] = {1.0f, 1.0595f, 1.1225f, 1.1892f, 1.2599f, 1.3348f, 1.4142f,
                                    1.4983f, 1.5874f, 1.6818f, 1.7818f, 1.8877f, 2.0f};
    if (n < 2 || f[0] == 0.0f)
        return {"Per8", 0.0f};
    float ratio = f[1] / f[0];
    while (ratio > 2.0f)
        ratio /= 2.0f;
    const float *table = system == AA_TUNING_JUST_INTONATION ? just : (system == AA_TUNING_PYTHAGOREAN ? pyth : equal);
    int nearest = 0;
    int32_t nearest_key = total_order_key(std::fabs(ratio - table[0]));
    for (int i = 1; i < 13; ++i) {
        const int32_t key = total_order_key(std::fabs(ratio - table[i]));
        if (key < nearest_key) {
            nearest = i;
            nearest_key = key;
        }
    }
    return {names[nearest], -std::log(table[nearest] / ratio) * 1732.5f};
}

// ---- JSON text as serde_json writes it ----------------------------------------------------
// Floats: shortest digits that round-trip, laid out like the `ryu` crate's pretty printer
// (serde_json's float writer): plain decimals for 1e-5 <= |v| < 1e16 (f32: 1e-6 .. 1e13, the bounds
// from memory of ryu/src/pretty), otherwise d.ddde[-]x; integers keep a ".0"; non-finite -> null.
template <typename T>
void put_float(std::string &js, T v, int plain_max, int plain_min)
{
    if (!std::isfinite(v)) {
        js += "null";
        return;
    }
    if (std::signbit(v)) {
        js += '-';
        v = -v;
    }
    if (v == 0) {
        js += "0.0";
        return;
    }
    char sci[48];
    auto r = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);
    *r.ptr = 0;
    // sci = d[.ddd]e[+-]xx
    std::string digits;
    const char *p = sci;
    for (; *p && *p != 'e'; ++p)
        if (*p != '.')
            digits += *p;
    const int e10 = std::atoi(p + 1);
    const int len = (int)digits.size();
    const int point = e10 + 1;        // 10^(point-1) <= v < 10^point
    const int trailing = point - len; // power of ten of the last digit
    if (trailing >= 0 && point <= plain_max) {
        js += digits;
        js.append((size_t)trailing, '0');
        js += ".0";
    } else if (point > 0 && point <= plain_max) {
        js.append(digits, 0, (size_t)point);
        js += '.';
        js.append(digits, (size_t)point, std::string::npos);
    } else if (point > plain_min && point <= 0) {
        js += "0.";
        js.append((size_t)-point, '0');
        js += digits;
    } else {
        js += digits[0];
        if (len > 1) {
            js += '.';
            js.append(digits, 1, std::string::npos);
        }
        js += 'e';
        js += std::to_string(point - 1);
    }
}

inline void put_f32(std::string &js, float v) { put_float<float>(js, v, 13, -6); }
inline void put_f64(std::string &js, double v) { put_float<double>(js, v, 16, -5); }

void put_string(std::string &js, const char *s)
{
    js += '"';
    for (const unsigned char *p = (const unsigned char *)s; *p; ++p) {
        switch (*p) {
        case '"': js += "\\\""; break;
        case '\\': js += "\\\\"; break;
        case '\b': js += "\\b"; break;
        case '\f': js += "\\f"; break;
        case '\n': js += "\\n"; break;
        case '\r': js += "\\r"; break;
        case '\t': js += "\\t"; break;
        default:
            if (*p < 0x20) {
                char u[8];
                std::snprintf(u, sizeof u, "\\u%04x", (unsigned)*p);
                js += u;
            } else {
                js += (char)*p;
            }
        }
    }
    js += '"';
}

}  // namespace

extern "C" {

int aa_note_from_freq(float freq, float base_freq, char *name, float *cents)
{
    if (!name || !cents)
        return aa::abi_fail(AA_ERR_ARG, "aa_note_from_freq: name or cents is NULL");
    const NoteId id = identify(freq, base_freq);
    write_name(id, name);
    *cents = id.cents;
    return AA_OK;
}

int aa_note_to_freq(const char *name, float cents, float base_freq, float *freq)
{
    if (!name || !freq)
        return aa::abi_fail(AA_ERR_ARG, "aa_note_to_freq: name or freq is NULL");
    const size_t len = std::strlen(name);
    if (len < 2)
        return aa::abi_fail(AA_ERR_ARG, "note \"%s\": too short, expected a letter, an optional accidental and an octave (C#4, A4)", name);
    static const char letters[] = "CDEFGAB";
    static const int from_a4[] = {-9, -7, -5, -4, -2, 0, 2};
    const char *hit = std::strchr(letters, name[0]);
    if (!hit || !name[0])
        return aa::abi_fail(AA_ERR_ARG, "note \"%s\": invalid letter '%c', expected one of C D E F G A B", name, name[0]);
    int semitones = from_a4[hit - letters];
    size_t digits_at = 2;
    switch (name[1]) {
    case '#': semitones += 1; break;
    case 'b': semitones -= 1; break;
    case 'n': break;
    case 'x':
        if (len > 2)
            semitones += 2;
        else
            digits_at = 1;
        break;
    case 'B':
        if (len > 2)
            semitones -= 2;
        else
            digits_at = 1;
        break;
    default: digits_at = 1;
    }
    // the octave parses as Rust's u8: optional '+', decimal digits, at most 255
    const char *d = name + digits_at;
    if (*d == '+')
        ++d;
    unsigned octave = 0;
    bool ok = *d != 0;
    for (; ok && *d; ++d) {
        ok = *d >= '0' && *d <= '9';
        if (ok) {
            octave = octave * 10 + (unsigned)(*d - '0');
            ok = octave <= 255;
        }
    }
    if (!ok)
        return aa::abi_fail(AA_ERR_ARG, "note \"%s\": invalid octave \"%s\", expected a number like 4", name, name + digits_at);
    semitones += ((int)octave - 4) * 12;
    *freq = base_freq * std::pow(2.0f, ((float)semitones + cents / 100.0f) / 12.0f);
    return AA_OK;
}

int aa_midi_from_freq(float freq, float base_freq, int32_t *midi, float *cents)
{
    if (!midi || !cents)
        return aa::abi_fail(AA_ERR_ARG, "aa_midi_from_freq: midi or cents is NULL");
    const float pos = cents_from_c0(freq, base_freq);
    *cents = nearest_semitone_offset(pos);
    *midi = (int32_t)((saturate_u8(std::round(pos / 100.0f)) + 12u) & 0xffu);
    return AA_OK;
}

int aa_midi_to_freq(int32_t midi, float cents, float base_freq, float *freq)
{
    if (!freq || midi < 0 || midi > 255)
        return aa::abi_fail(AA_ERR_ARG, "aa_midi_to_freq: midi %d (0..255), freq %p", midi, (void *)freq);
    *freq = base_freq * std::pow(2.0f, ((float)midi - 69.0f + cents / 100.0f) / 12.0f);
    return AA_OK;
}

int aa_interval(const float *freqs, int32_t n, int32_t system, char *name, float *accuracy)
{
    if (!name || !accuracy || n < 0 || (n > 0 && !freqs))
        return aa::abi_fail(AA_ERR_ARG, "aa_interval: bad argument");
    if (n >= 2 && freqs[0] != 0.0f && freqs[1] / freqs[0] == INFINITY)
        return aa::abi_fail(AA_ERR_ARG, "aa_interval: infinite ratio (the reference's octave-folding loop would not end)");
    const IntervalId iv = classify_interval(freqs, n, system);
    std::snprintf(name, AA_NOTE_NAME_LEN, "%s", iv.name);
    *accuracy = iv.accuracy;
    return AA_OK;
}

int aa_tuner_output_build(const float *freq_host, const float *score_host, int32_t n, int32_t mode, int32_t system,
                          float base_freq, double beat_position, aa_tuner_output *out)
{
    if (!out || n < 0 || n > AA_TUNER_NOTES_MAX || (n > 0 && (!freq_host || !score_host)))
        return aa::abi_fail(AA_ERR_ARG, "aa_tuner_output_build: n=%d (0..%d) or NULL argument", n, AA_TUNER_NOTES_MAX);
    if (mode != AA_TUNER_MULTI_PITCH && mode != AA_TUNER_SINGLE_PITCH)
        return aa::abi_fail(AA_ERR_ARG, "aa_tuner_output_build: mode %d", mode);
    if (system < AA_TUNING_EQUAL_TEMPERAMENT || system > AA_TUNING_PYTHAGOREAN)
        return aa::abi_fail(AA_ERR_ARG, "aa_tuner_output_build: tuning system %d", system);
    std::memset(out, 0, sizeof *out);
    out->mode = mode;
    out->system = system;
    // TunerCommand::SetBaseFreq clamps (tuner.rs:119); 440 is the initial value
    const float a4 = std::fmin(std::fmax(base_freq, 220.0f), 880.0f);
    out->base_freq = a4;
    out->beat_position = beat_position;
    if (n == 0)
        return AA_OK;   // the reference skips empty lists and leaves the previous output (tuner.rs:149-151)
    if (n == 1 || mode == AA_TUNER_SINGLE_PITCH) {
        int best = 0;
        for (int i = 1; i < n; ++i)
            if (total_order_key(score_host[i]) >= total_order_key(score_host[best]))
                best = i;
        const NoteId id = identify(freq_host[best], a4);
        write_name(id, out->notes[0]);
        std::snprintf(out->label, sizeof out->label, "%s", out->notes[0]);
        out->cents = out->accuracies[0] = id.cents;
        out->n_notes = 1;
    } else if (n == 2) {
        float pair[2] = {freq_host[0], freq_host[1]};
        if (total_order_key(pair[1]) < total_order_key(pair[0])) {
            pair[0] = freq_host[1];
            pair[1] = freq_host[0];
        }
        if (pair[0] != 0.0f && pair[1] / pair[0] == INFINITY)
            return aa::abi_fail(AA_ERR_ARG, "aa_tuner_output_build: infinite frequency ratio");
        const IntervalId iv = classify_interval(pair, 2, system);
        for (int i = 0; i < 2; ++i) {
            const NoteId id = identify(pair[i], a4);
            write_name(id, out->notes[i]);
            out->accuracies[i] = id.cents;
        }
        std::snprintf(out->label, sizeof out->label, "%s", iv.name);
        out->cents = iv.accuracy;
        out->n_notes = 2;
    } else {
        std::string label;
        for (int i = 0; i < n; ++i) {
            const NoteId id = identify(freq_host[i], a4);
            write_name(id, out->notes[i]);
            out->accuracies[i] = id.cents;
            if (i)
                label += ' ';
            label += out->notes[i];
        }
        std::snprintf(out->label, sizeof out->label, "%s", label.c_str());
        out->n_notes = n;
    }
    return AA_OK;
}

int aa_tuner_output_json(const aa_tuner_output *out, const char *key, char *buf, size_t cap, size_t *len)
{
    if (!out || !key || !len || (cap > 0 && !buf) || out->n_notes < 0 || out->n_notes > AA_TUNER_NOTES_MAX)
        return aa::abi_fail(AA_ERR_ARG, "aa_tuner_output_json: bad argument");
    static const char *const modes[2] = {"MultiPitch", "SinglePitch"};
    static const char *const systems[3] = {"EqualTemperament", "JustIntonation", "Pythagorean"};
    if (out->mode < 0 || out->mode > 1 || out->system < 0 || out->system > 2)
        return aa::abi_fail(AA_ERR_ARG, "aa_tuner_output_json: mode %d / system %d", out->mode, out->system);
    // field order of `struct TunerOutput` (tuner.rs:38-56)
    std::string js = "{\"label\":";
    put_string(js, out->label);
    js += ",\"cents\":";
    put_f32(js, out->cents);
    js += ",\"notes\":[";
    for (int i = 0; i < out->n_notes; ++i) {
        if (i)
            js += ',';
        put_string(js, out->notes[i]);
    }
    js += "],\"accuracies\":[";
    for (int i = 0; i < out->n_notes; ++i) {
        if (i)
            js += ',';
        put_f32(js, out->accuracies[i]);
    }
    js += "],\"mode\":\"";
    js += modes[out->mode];
    js += "\",\"system\":\"";
    js += systems[out->system];
    js += "\",\"base_freq\":";
    put_f32(js, out->base_freq);
    js += ",\"key\":";
    put_string(js, key);
    js += ",\"beat_position\":";
    put_f64(js, out->beat_position);
    js += '}';
    *len = js.size();
    if (js.size() + 1 > cap)
        return aa::abi_fail(AA_ERR_LEN, "aa_tuner_output_json: needs %zu bytes, buffer has %zu", js.size() + 1, cap);
    std::memcpy(buf, js.c_str(), js.size() + 1);
    return AA_OK;
}

}  // extern "C"

// Offline analyser for a recorded practice session, written against the C ABI only
// (include/aa_stft.h + the HIP runtime for device buffers): what a compiled host — the reference is
// a Rust crate — does with libaa_stft.so.  Reads the recorder's WAV (16-bit PCM mono,
// src/audio_io/recorder.rs:60-85), runs the tuner chain (2048/512 STFT -> noise floor -> pitches ->
// PitchTracker, src/audio_io/stft.rs) and the onset chain (256/64 STFT -> flux -> gates,
// src/analysis/onset.rs) on the GPU and prints JSON:
//   {"sample_rate":..,"frames":..,"onsets":[{"beat_position":..,"raw_sample_offset":..,"velocity":..},..],
//    "notes":[{"frame":..,"pitches":[[freq,score],..]},..]}
// The samples first pass the reducer thread's conditioning (aa_conditioner_*: biquads, gate, AGC;
// audio_io/mod.rs:351-478), which also yields noise_floor_db per 1024-sample slot - the
// detectors' global_floor - exactly as the live engine feeds its workers.
// Usage: aa_analyze <file.wav> [bpm=120] [note_stride=50]
// Build: make -C audio_analyzer_rs_amd/csrc examples
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "aa_stft.h"

#define CHECK_AA(call)                                                                      \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != AA_OK) {                                                                 \
            std::fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, aa_last_error());      \
            return 2;                                                                       \
        }                                                                                   \
    } while (0)
#define CHECK_HIP(call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));          \
            return 2;                                                                       \
        }                                                                                   \
    } while (0)

// Minimal RIFF/WAVE reader: 16-bit PCM, first channel pair averaged like the capture path
// (audio_io/mod.rs:784-794); samples decoded as i16 / 32768.
static bool read_wav(const char *path, std::vector<float> *out, int *sample_rate)
{
    FILE *f = std::fopen(path, "rb");
    if (!f)
        return false;
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0)
        buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    if (buf.size() < 44 || std::memcmp(buf.data(), "RIFF", 4) != 0 || std::memcmp(buf.data() + 8, "WAVE", 4) != 0)
        return false;
    auto u16 = [&](size_t o) { return (unsigned)buf[o] | ((unsigned)buf[o + 1] << 8); };
    auto u32 = [&](size_t o) { return (uint32_t)u16(o) | ((uint32_t)u16(o + 2) << 16); };
    int channels = 0, bits = 0, fmt = 0;
    size_t pos = 12;
    while (pos + 8 <= buf.size()) {
        const uint32_t len = u32(pos + 4);
        if (std::memcmp(buf.data() + pos, "fmt ", 4) == 0 && pos + 24 <= buf.size()) {
            fmt = (int)u16(pos + 8);
            channels = (int)u16(pos + 10);
            *sample_rate = (int)u32(pos + 12);
            bits = (int)u16(pos + 22);
        } else if (std::memcmp(buf.data() + pos, "data", 4) == 0) {
            if (fmt != 1 || bits != 16 || channels < 1)
                return false;
            const size_t avail = std::min<size_t>(len, buf.size() - pos - 8);
            const size_t frames = avail / (2 * (size_t)channels);
            const int used = channels < 2 ? 1 : 2;
            out->resize(frames);
            const unsigned char *pcm = buf.data() + pos + 8;           // little-endian int16, interleaved
            auto s16 = [&](size_t idx) { return (float)(int16_t)((uint16_t)pcm[2 * idx] | ((uint16_t)pcm[2 * idx + 1] << 8)) / 32768.0f; };
            if (channels == 1) {
                for (size_t i = 0; i < frames; ++i)
                    (*out)[i] = (0.0f + s16(i)) / 1.0f;
            } else {
                for (size_t i = 0; i < frames; ++i) {
                    float acc = 0.0f;
                    for (int c = 0; c < used; ++c)
                        acc += s16(i * (size_t)channels + (size_t)c);
                    (*out)[i] = acc / (float)used;
                }
            }
            return true;
        }
        pos += 8 + len + (len & 1);
    }
    return false;
}

// wall-clock phase marks on stderr when AA_ANALYZE_TIMING is set
struct PhaseClock {
    bool on = std::getenv("AA_ANALYZE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        const auto now = std::chrono::steady_clock::now();
        if (on)
            std::fprintf(stderr, "%-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int main(int argc, char **argv)
{
    PhaseClock phase;
    // the HIP runtime takes ~0.1 s to come up: let it do so next to the file read and the host-side conditioning
    struct Joined {
        std::thread t;
        ~Joined()
        {
            if (t.joinable())
                t.join();
        }
    } hip_up{std::thread([] { (void)hipFree(nullptr); })};
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s file.wav [bpm=120] [note_stride=50]\n", argv[0]);
        return 1;
    }
    const double bpm = argc > 2 ? std::atof(argv[2]) : 120.0;
    const int note_stride = argc > 3 ? std::atoi(argv[3]) : 50;
    std::vector<float> x;
    int sr = 0;
    if (!read_wav(argv[1], &x, &sr)) {
        std::fprintf(stderr, "%s: not a 16-bit PCM WAV file\n", argv[1]);
        return 1;
    }
    phase.mark("read wav");
    // ---- conditioning: whole 1024-sample slots, in place; noise floor per slot ----
    const int slot_len = 1024;
    const int64_t n_slots = (int64_t)x.size() / slot_len;
    x.resize((size_t)(n_slots * slot_len));
    std::vector<float> nf_db((size_t)std::max<int64_t>(n_slots, 1), -60.0f), rms_db(nf_db.size()), gain_db(nf_db.size()),
        med_db(nf_db.size());
    std::vector<int32_t> level(nf_db.size());
    aa_conditioner *cond = nullptr;
    CHECK_AA(aa_conditioner_create((float)sr, slot_len, &cond));
    CHECK_AA(aa_conditioner_process(cond, x.data(), n_slots, nf_db.data(), rms_db.data(), gain_db.data(),
                                    med_db.data(), level.data()));
    aa_conditioner_destroy(cond);
    phase.mark("conditioner (host)");
    // global_floor of frame t: the slot holding its last sample (stft.rs:322-324, onset.rs:299-300)
    auto global_floor = [&](int64_t frames, int n_fft, int hop, int n_bins) {
        std::vector<float> gf((size_t)frames);
        for (int64_t t = 0; t < frames; ++t) {
            int64_t slot = (t * hop + n_fft - 1) / slot_len;
            slot = std::min<int64_t>(slot, n_slots - 1);
            gf[(size_t)t] = std::pow(10.0f, nf_db[(size_t)slot] / 20.0f) * (float)n_bins / 2.0f;
        }
        return gf;
    };
    const int64_t n = (int64_t)x.size();
    hip_up.t.join();
    float *d_x = nullptr;
    CHECK_HIP(hipMalloc(&d_x, sizeof(float) * (size_t)std::max<int64_t>(n, 1)));
    CHECK_HIP(hipMemcpy(d_x, x.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    phase.mark("upload (HIP came up in the background)");

    // ---- tuner chain: 2048 / 512 (stft.rs:169-170) ----
    aa_stft_config cfg{};
    cfg.n_fft = 2048;
    cfg.hop = 512;
    cfg.window = AA_WINDOW_HANN_PERIODIC;
    cfg.output = AA_OUT_MAG;
    cfg.sample_rate = sr;
    aa_stft_plan *tuner = nullptr;
    CHECK_AA(aa_stft_plan_create(&cfg, nullptr, &tuner));
    const int64_t tf = aa_stft_num_frames(tuner, n);
    const int tb = aa_stft_out_cols(tuner);
    std::vector<float> nf_freq((size_t)tf * 8), nf_score((size_t)tf * 8);
    std::vector<int32_t> nf_count((size_t)tf);
    std::vector<float> st_freq((size_t)tf * AA_PITCH_TRACKS_MAX), st_score((size_t)tf * AA_PITCH_TRACKS_MAX);
    std::vector<int32_t> st_count((size_t)tf);
    if (tf > 0) {
        float *d_mag, *d_eff, *d_gf, *d_state, *d_freq, *d_score;
        int32_t *d_count;
        CHECK_HIP(hipMalloc(&d_mag, sizeof(float) * (size_t)tf * tb));
        CHECK_HIP(hipMalloc(&d_eff, sizeof(float) * (size_t)tf * tb));
        CHECK_HIP(hipMalloc(&d_gf, sizeof(float) * (size_t)tf));
        CHECK_HIP(hipMalloc(&d_state, sizeof(float) * 3 * (size_t)tb));
        CHECK_HIP(hipMalloc(&d_freq, sizeof(float) * (size_t)tf * 8));
        CHECK_HIP(hipMalloc(&d_score, sizeof(float) * (size_t)tf * 8));
        CHECK_HIP(hipMalloc(&d_count, sizeof(int32_t) * (size_t)tf));
        std::vector<float> gf = global_floor(tf, 2048, 512, tb);
        CHECK_HIP(hipMemcpy(d_gf, gf.data(), sizeof(float) * (size_t)tf, hipMemcpyHostToDevice));
        CHECK_AA(aa_stft_execute_device(tuner, d_x, n, 0, tf, d_mag, nullptr));
        CHECK_AA(aa_tuner_floor_device(d_mag, tf, tb, d_gf, d_state, 0, d_eff, nullptr));
        CHECK_AA(aa_tuner_pitches_device(d_mag, d_eff, tf, tb, (float)sr / 2048.0f, 24.0f, 10000.0f, d_freq,
                                         d_score, d_count, nullptr));
        CHECK_HIP(hipMemcpy(nf_freq.data(), d_freq, sizeof(float) * (size_t)tf * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(nf_score.data(), d_score, sizeof(float) * (size_t)tf * 8, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(nf_count.data(), d_count, sizeof(int32_t) * (size_t)tf, hipMemcpyDeviceToHost));
        aa_pitch_tracker *trk = nullptr;
        CHECK_AA(aa_pitch_tracker_create(&trk));
        CHECK_AA(aa_pitch_tracker_process(trk, nf_freq.data(), nf_score.data(), nf_count.data(), nullptr, tf,
                                          st_freq.data(), st_score.data(), st_count.data()));
        aa_pitch_tracker_destroy(trk);
        for (void *p : {(void *)d_mag, (void *)d_eff, (void *)d_gf, (void *)d_state, (void *)d_freq, (void *)d_score,
                        (void *)d_count})
            (void)hipFree(p);
    }
    aa_stft_plan_destroy(tuner);
    phase.mark("tuner chain");

    // ---- onset chain: 256 / 64 (onset.rs:122-123) ----
    cfg.n_fft = 256;
    cfg.hop = 64;
    aa_stft_plan *onset = nullptr;
    CHECK_AA(aa_stft_plan_create(&cfg, nullptr, &onset));
    const int64_t of = aa_stft_num_frames(onset, n);
    const int ob = aa_stft_out_cols(onset);
    std::vector<uint8_t> flags((size_t)of);
    std::vector<float> velocity((size_t)of);
    if (of > 0) {
        float *d_mag, *d_flux, *d_energy, *d_gf, *d_state, *d_oflux, *d_ema, *d_exc, *d_vel;
        int32_t *d_cnt;
        uint8_t *d_flags;
        CHECK_HIP(hipMalloc(&d_mag, sizeof(float) * (size_t)of * ob));
        for (float **p : {&d_flux, &d_energy, &d_gf, &d_oflux, &d_ema, &d_exc, &d_vel})
            CHECK_HIP(hipMalloc(p, sizeof(float) * (size_t)of));
        CHECK_HIP(hipMalloc(&d_state, sizeof(float) * (size_t)(ob + 4)));
        CHECK_HIP(hipMalloc(&d_cnt, sizeof(int32_t) * (size_t)of));
        CHECK_HIP(hipMalloc(&d_flags, (size_t)of));
        std::vector<float> gf = global_floor(of, 256, 64, ob);
        CHECK_HIP(hipMemcpy(d_gf, gf.data(), sizeof(float) * (size_t)of, hipMemcpyHostToDevice));
        CHECK_AA(aa_stft_execute_device(onset, d_x, n, 0, of, d_mag, nullptr));
        CHECK_AA(aa_onset_flux_device(d_mag, of, ob, nullptr, d_flux, d_energy, nullptr));
        CHECK_AA(aa_onset_gates_device(d_mag, of, ob, d_flux, d_energy, d_gf, nullptr, 1, d_state, 0, d_oflux, d_ema, d_exc,
                                       d_cnt, d_flags, d_vel, nullptr));
        CHECK_HIP(hipMemcpy(flags.data(), d_flags, (size_t)of, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(velocity.data(), d_vel, sizeof(float) * (size_t)of, hipMemcpyDeviceToHost));
        for (void *p : {(void *)d_mag, (void *)d_flux, (void *)d_energy, (void *)d_gf, (void *)d_state,
                        (void *)d_oflux, (void *)d_ema, (void *)d_exc, (void *)d_vel, (void *)d_cnt, (void *)d_flags})
            (void)hipFree(p);
    }
    aa_stft_plan_destroy(onset);
    (void)hipFree(d_x);

    phase.mark("onset chain");
    // ---- JSON ----
    std::string js = "{\"sample_rate\":" + std::to_string(sr) + ",\"frames\":" + std::to_string(tf) + ",\"onsets\":[";
    bool first = true;
    char item[160];
    for (int64_t t = 0; t < of; ++t) {
        if (!(flags[(size_t)t] & 16))
            continue;
        // window centre of frame t; the format of OnsetDetection::poll_onsets (lib.rs:326-338)
        const int64_t centre = t * 64 + 128;
        std::snprintf(item, sizeof item, "%s{\"beat_position\":%.6f,\"raw_sample_offset\":%lld,\"velocity\":%.4f}",
                      first ? "" : ",", (double)centre / sr * bpm / 60.0, (long long)centre, velocity[(size_t)t]);
        js += item;
        first = false;
    }
    js += "],\"notes\":[";
    first = true;
    for (int64_t t = 0; t < tf; t += note_stride > 0 ? note_stride : 1) {
        std::snprintf(item, sizeof item, "%s{\"frame\":%lld,\"pitches\":[", first ? "" : ",", (long long)t);
        js += item;
        for (int i = 0; i < st_count[(size_t)t]; ++i) {
            std::snprintf(item, sizeof item, "%s[%.3f,%.4f]", i ? "," : "",
                          st_freq[(size_t)t * AA_PITCH_TRACKS_MAX + i], st_score[(size_t)t * AA_PITCH_TRACKS_MAX + i]);
            js += item;
        }
        js += "]";
        // what Tuner::poll_output would have shown after this frame (tuner.rs:141-207, lib.rs:80-82)
        aa_tuner_output tout;
        const int64_t centre = t * 512 + 1024;
        CHECK_AA(aa_tuner_output_build(&st_freq[(size_t)t * AA_PITCH_TRACKS_MAX], &st_score[(size_t)t * AA_PITCH_TRACKS_MAX],
                                       st_count[(size_t)t], AA_TUNER_MULTI_PITCH, AA_TUNING_EQUAL_TEMPERAMENT, 440.0f,
                                       (double)centre / sr * bpm / 60.0, &tout));
        if (tout.n_notes > 0) {
            char tj[2048];
            size_t tl = 0;
            CHECK_AA(aa_tuner_output_json(&tout, "C major", tj, sizeof tj, &tl));
            js += ",\"tuner\":";
            js += tj;
        }
        js += "}";
        first = false;
    }
    js += "]}";
    std::puts(js.c_str());
    phase.mark("JSON");
    return 0;
}

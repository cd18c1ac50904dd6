// Sanitizer fuzz of the host-only TunerOutput entry points (aa_theory.cpp): CPU build, ASAN + UBSAN
// (GPU sanitizers are not available on the pool).  Random bit patterns for every float argument, random
// short strings for note names, pitch lists of every length; checks bounds of the name / JSON buffers.
// Build + run (about 40 s):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ \
//       -I/opt/rocm/include -Wno-unknown-pragmas -o tools/bin/fuzz_theory tools/fuzz_theory.cpp \
//       audio_analyzer_rs_amd/csrc/aa_theory.cpp && tools/bin/fuzz_theory
#include "../include/aa_stft.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
namespace aa { int abi_fail(int code, const char *, ...) { return code; } }
int main()
{
    std::mt19937 rng(7);
    std::uniform_int_distribution<uint32_t> bits;
    auto rf = [&]() { uint32_t b = bits(rng); float f; std::memcpy(&f, &b, 4); return f; };
    char name[AA_NOTE_NAME_LEN];
    float c, f;
    int32_t midi;
    long n_ok = 0;
    for (int i = 0; i < 2000000; ++i) {
        float x = (i & 1) ? rf() : 20.0f * std::exp((bits(rng) % 7000) * 0.001f), base = (i & 2) ? rf() : 440.0f;
        n_ok += aa_note_from_freq(x, base, name, &c) == AA_OK;
        if (std::strlen(name) >= AA_NOTE_NAME_LEN) return 1;
        aa_note_to_freq(name, c, base, &f);
        aa_midi_from_freq(x, base, &midi, &c);
        if (midi < 0 || midi > 255) return 2;
        aa_midi_to_freq(midi, c, base, &f);
        float pr[2] = {rf(), x};
        char iv[AA_NOTE_NAME_LEN];
        aa_interval(pr, bits(rng) % 3, bits(rng) % 3, iv, &c);
        float pr2[2] = {x, x * (1.0f + (bits(rng) % 1000) * 0.004f)};
        aa_interval(pr2, 2, bits(rng) % 3, iv, &c);
    }
    // note names: random short strings over a small alphabet
    const char alpha[] = "ABCDEFGHXabnx#+-0123456789 .";
    for (int i = 0; i < 2000000; ++i) {
        char s[8];
        int len = bits(rng) % 7;
        for (int k = 0; k < len; ++k) s[k] = alpha[bits(rng) % (sizeof alpha - 1)];
        s[len] = 0;
        aa_note_to_freq(s, 0.0f, 440.0f, &f);
    }
    // tuner output + JSON with arbitrary bit patterns
    aa_tuner_output out;
    char js[4096];
    size_t len;
    for (int i = 0; i < 300000; ++i) {
        float fr[AA_TUNER_NOTES_MAX], sc[AA_TUNER_NOTES_MAX];
        int n = bits(rng) % (AA_TUNER_NOTES_MAX + 1);
        for (int k = 0; k < n; ++k) { fr[k] = (i & 1) ? rf() : 50.0f + bits(rng) % 4000; sc[k] = (i & 4) ? rf() : (float)(bits(rng) % 5); }
        double beat; uint64_t bb = ((uint64_t)bits(rng) << 32) | bits(rng); std::memcpy(&beat, &bb, 8);
        int rc = aa_tuner_output_build(fr, sc, n, bits(rng) % 2, bits(rng) % 3, (i & 8) ? rf() : 440.0f, beat, &out);
        if (rc == AA_OK) {
            if (aa_tuner_output_json(&out, "C \"major\"\n", js, sizeof js, &len) != AA_OK) return 3;
            if (std::strlen(js) != len) return 4;
            aa_tuner_output_json(&out, "k", js, 16, &len);     // too small: must not write past 16
        }
    }
    std::printf("fuzz ok (%ld names)\n", n_ok);
    return 0;
}

// C-ABI layer of libaa_stft.so: plans, tables, workspaces, error reporting.
// Public contract and the reference lines each entry replaces: include/aa_stft.h.
#include "../../include/aa_stft.h"
#include "aa_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {
thread_local char g_err[512] = "";
}

// records the message aa_last_error() returns (shared with aa_theory.cpp) and hands the code back
int aa::abi_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define fail(...) aa::abi_fail(__VA_ARGS__)

namespace {

#define AA_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(AA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                 \
    } while (0)

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

constexpr double kPi = 3.14159265358979323846264338327950288;

using aa::DeviceGuard;

// Twiddle tables shared by the generic kernels: W_m^q (q < m/2 for powers of two,
// q < m otherwise) and the split table W_n^k, k <= m/2.  f64 evaluation, f32 storage.
struct FftTables {
    int n = 0;
    bool pow2 = false;
    float *tw_m = nullptr;      // device
    float *tw_split = nullptr;  // device
};

int build_fft_tables(int n, FftTables *t)
{
    const int m = n / 2;
    t->n = n;
    t->pow2 = is_pow2(n);
    const int ntw = m;   // W_m^q, q < m (the Stockham passes index up to (R-1)/R of the circle)
    std::vector<float> a(2 * (size_t)ntw), b(2 * (size_t)(m / 2 + 1));
    for (int q = 0; q < ntw; ++q) {
        double ang = -2.0 * kPi * (double)q / (double)m;
        a[2 * q] = (float)std::cos(ang);
        a[2 * q + 1] = (float)std::sin(ang);
    }
    for (int k = 0; k <= m / 2; ++k) {
        double ang = -2.0 * kPi * (double)k / (double)n;
        b[2 * k] = (float)std::cos(ang);
        b[2 * k + 1] = (float)std::sin(ang);
    }
    AA_HIP(hipMalloc(&t->tw_m, a.size() * sizeof(float)));
    AA_HIP(hipMalloc(&t->tw_split, b.size() * sizeof(float)));
    AA_HIP(hipMemcpy(t->tw_m, a.data(), a.size() * sizeof(float), hipMemcpyHostToDevice));
    AA_HIP(hipMemcpy(t->tw_split, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    return AA_OK;
}

void free_fft_tables(FftTables *t)
{
    if (t->tw_m)
        (void)hipFree(t->tw_m);
    if (t->tw_split)
        (void)hipFree(t->tw_split);
    t->tw_m = t->tw_split = nullptr;
}

int check_len(int len)
{
    if (len < 4 || (len & 1))
        return fail(AA_ERR_LEN, "FFT length %d: must be even and >= 4", len);
    if (len > 8192)
        return fail(AA_ERR_LEN, "FFT length %d: this build supports lengths up to 8192", len);
    return AA_OK;
}

void hann_f32(int n, float *w)
{
    // src/audio_io/stft.rs:641-648 — all-f32 periodic Hann.
    const float pi_f32 = 3.14159265358979323846f;
    for (int i = 0; i < n; ++i) {
        float x = (float)i / (float)n;
        float arg = 2.0f * pi_f32 * x;
        w[i] = 0.5f - 0.5f * cosf(arg);
    }
}

// Per-lane constants of the wave-per-frame 1024 kernel, [kWave1024Rows][64]:
//   rows  0..15  0.5 * w[2*(64 j + lane) + c]         (row 2j + c)
//   rows 16..29  W_512^(lane k)  re,im   k = 1..7      (pass A twiddles)
//   rows 30..43  W_64^(b k)    re,im   k = 1..7, b = lane>>3   (pass B twiddles)
//   rows 44..51  (-sin t, -cos t), t = 2 pi (lane + 64 j)/1024, j = 0..3   (= -i W_1024^k)
void build_wave1024_table(const float *window, std::vector<float> *tab)
{
    tab->assign((size_t)aa::kWave1024Rows * 64, 0.0f);
    float *T = tab->data();
    for (int lane = 0; lane < 64; ++lane) {
        const int n2 = lane;   // pass A reads complex point m = 64 j + lane
        const int b = lane >> 3;
        for (int j = 0; j < 8; ++j)
            for (int c = 0; c < 2; ++c)
                T[(2 * j + c) * 64 + lane] = 0.5f * window[2 * (64 * j + n2) + c];
        for (int k = 1; k < 8; ++k) {
            double aA = -2.0 * kPi * (double)(n2 * k) / 512.0;
            T[(16 + 2 * (k - 1)) * 64 + lane] = (float)std::cos(aA);
            T[(17 + 2 * (k - 1)) * 64 + lane] = (float)std::sin(aA);
            double aB = -2.0 * kPi * (double)(b * k) / 64.0;
            T[(30 + 2 * (k - 1)) * 64 + lane] = (float)std::cos(aB);
            T[(31 + 2 * (k - 1)) * 64 + lane] = (float)std::sin(aB);
        }
        for (int j = 0; j < 4; ++j) {
            double th = 2.0 * kPi * (double)(lane + 64 * j) / 1024.0;
            T[(44 + 2 * j) * 64 + lane] = (float)(-std::sin(th));
            T[(45 + 2 * j) * 64 + lane] = (float)(-std::cos(th));
        }
    }
}

constexpr int64_t kMfccSlab = 262144; // frames per log-mel slab of the fused MFCC path
constexpr int64_t kMelSlab = 32768;  // frames per power-spectrogram slab (<= 134 MB: Infinity-Cache resident; 256 GEMM workgroups)

// Slaney mel scale (linear below 1 kHz, log above) — the build's definition of AA_OUT_LOGMEL
// (the reference has no mel stage); oracle/truth.py holds the float64 twin.
double hz_to_mel(double f)
{
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m)
{
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}
// Slaney-normalised triangular filters, row-major [n_mels][n_bins], float64.
void mel_filterbank(int sr, int n_fft, int n_mels, double fmin, double fmax, std::vector<double> *w)
{
    const int nb = n_fft / 2 + 1;
    w->assign((size_t)n_mels * nb, 0.0);
    std::vector<double> hz(n_mels + 2);
    const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i)
        hz[i] = mel_to_hz(m0 + (m1 - m0) * (double)i / (double)(n_mels + 1));
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (hz[i + 2] - hz[i]);
        for (int k = 0; k < nb; ++k) {
            const double f = (double)k * sr / n_fft;
            const double lower = (f - hz[i]) / (hz[i + 1] - hz[i]);
            const double upper = (hz[i + 2] - f) / (hz[i + 2] - hz[i + 1]);
            const double v = std::min(lower, upper);
            (*w)[(size_t)i * nb + k] = v > 0.0 ? v * enorm : 0.0;
        }
    }
}

// Cuts n_mels filters of widths len[] into <= kMelMaxGroups groups of (lanes >> pl) consecutive
// mels with 1 << pl K-slices each (pl = 0, 1, 2): a small dynamic programme minimising
// sum(bins per slice + per-group overhead).  Fills n_mels / n_groups / trips / mel_base /
// part_log2 of *sch.  overhead_bins: cost of one more group in units of bins per slice.
bool plan_mel_groups(const std::vector<int> &len, int n_mels, int lanes, int nb, int overhead_bins,
                     aa::MelSched *sch)
{
    auto trips_of = [&](int first, int pl) {
        int widest = 0;
        for (int m = first; m < std::min(n_mels, first + (lanes >> pl)); ++m)
            widest = std::max(widest, len[m]);
        return ((widest + (1 << pl) - 1) / (1 << pl) + 3) / 4 * 4;   // the kernels step 4 bins at a time
    };
    // best[j][m]: cheapest way to cover mels [m, n_mels) with exactly j more groups
    const int G = aa::kMelMaxGroups, kOverhead = overhead_bins, kInf = 1 << 28;
    std::vector<std::vector<int>> best(G + 1, std::vector<int>(n_mels + 1, kInf)), pick = best;
    best[0][n_mels] = 0;
    for (int j = 1; j <= G; ++j)
        for (int m = n_mels - 1; m >= 0; --m)
            for (int pl = 0; pl <= 2; ++pl) {
                const int next = std::min(n_mels, m + (lanes >> pl));
                const int t = trips_of(m, pl);
                if ((t << pl) > nb || best[j - 1][next] >= kInf)
                    continue;
                const int cost = best[j - 1][next] + std::max(t, 4) + kOverhead;
                if (cost < best[j][m]) {
                    best[j][m] = cost;
                    pick[j][m] = pl;
                }
            }
    int jbest = 0;
    for (int j = 1; j <= G; ++j)
        if (best[j][0] < (jbest ? best[jbest][0] : kInf))
            jbest = j;
    if (!jbest)
        return false;
    *sch = aa::MelSched();
    sch->n_mels = n_mels;
    sch->n_groups = jbest;
    for (int g = 0, m = 0, j = jbest; j > 0; ++g, --j) {
        const int pl = pick[j][m];
        sch->trips[g] = std::max(trips_of(m, pl), 4);
        sch->mel_base[g] = m;
        sch->part_log2[g] = pl;
        m = std::min(n_mels, m + (lanes >> pl));
    }
    return true;
}

// Parameters and schedule of the fused mel reduction (aa_internal.hpp: MelSched): the same Slaney
// triangles as mel_filterbank, as two lines per filter in bin units.  `lanes` lanes reduce one
// frame (64 at n_fft 2048, 32 at 1024).  The mels are cut into <= kMelMaxGroups groups of
// lanes >> pl consecutive mels (pl = 0, 1, 2: 1, 2 or 4 K-slices per filter); a small dynamic
// programme picks the cut that minimises sum(trips + per-group overhead).  k_start is pulled
// down so that every slice stays inside the row (weights below the filter's first bin evaluate
// to 0 through med3).
bool build_mel_sched(int sr, int n_fft, int n_mels, double fmin, double fmax, int lanes,
                     std::vector<float> *par, aa::MelSched *sch)
{
    const int nb = n_fft / 2 + 1;
    if (n_mels < 1 || n_mels > aa::kMelMaxMels)
        return false;
    std::vector<double> c(n_mels + 2), hz(n_mels + 2);
    const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i) {
        hz[i] = mel_to_hz(m0 + (m1 - m0) * (double)i / (double)(n_mels + 1));
        c[i] = hz[i] * n_fft / sr;
    }
    std::vector<int> k_lo(n_mels), len(n_mels);
    for (int m = 0; m < n_mels; ++m) {
        k_lo[m] = std::max(0, (int)std::floor(c[m]) + 1);                     // first k > c[m]
        const int k_hi = std::min(nb - 1, (int)std::ceil(c[m + 2]) - 1);      // last  k < c[m+2]
        len[m] = std::max(0, k_hi - k_lo[m] + 1);
    }
    if (!plan_mel_groups(len, n_mels, lanes, nb, 8, sch))
        return false;
    par->assign(aa::kMelParTableDwords, 0.0f);
    for (int g = 0; g < sch->n_groups; ++g) {
        const int m = sch->mel_base[g], pl = sch->part_log2[g], parts = 1 << pl, trips = sch->trips[g];
        const int next = std::min(n_mels, m + (lanes >> pl));
        for (int i = m; i < next; ++i) {
            const int ks = std::max(0, std::min(k_lo[i], nb - parts * trips));
            const double e = 2.0 / (hz[i + 2] - hz[i]);
            const double iu = 1.0 / (c[i + 1] - c[i]), id = 1.0 / (c[i + 2] - c[i + 1]);
            float *q = par->data() + (size_t)aa::kMelParDwords * i;
            std::memcpy(&q[0], &ks, sizeof(int));
            q[1] = (float)(e * ((double)ks - c[i]) * iu);
            q[2] = (float)(e * iu);
            q[3] = (float)(e * (c[i + 2] - (double)ks) * id);
            q[4] = (float)(e * id);
        }
    }
    return true;
}

// kLogMelM / kMfccM (aa_internal.hpp, "band-blocked mel contraction"): the program of the four waves of a
// workgroup.  The filters are taken in quads of 4 consecutive mels; quad q touches the 4-bin pieces
// [c_lo, c_hi) of a power row (the last piece of a row - bin nb - 1 = Nyquist and three floats the kernel keeps
// at zero behind it - is a piece like any other).  With S steps per block a quad needs ceil(pieces / S) blocks
// (K-slices, consecutive blocks of ONE wave); a wave offers 16 blocks and 8 emission slots (quad slot x 8
// frames = its 64 lanes).  The quads go to the waves largest first, each to the wave with the most free blocks,
// and the smallest S that packs wins (fewer steps = fewer MFMAs and fewer resident registers).  The table is
// verified against the dense filterbank before it is accepted: the kernel's dataflow run on the table must
// reproduce every f32 weight, each exactly once.
// n_fft 1024: S_MAX = kMelMfmaSteps, 8 frames per emission slot (8 slots per wave), rows of kMelMfmaRowB floats;
// n_fft 2048: S_MAX = kMelMfmaSteps2048, steps_gated (the kernel multiplies n_steps steps only), 8 lanes per slot too (4 frames
// x the two halves of the quad).
bool build_mel_mfma(const std::vector<double> &w, int nb, int n_mels, int S_MAX, int frames_per_slot, int row_floats,
                    bool steps_gated, std::vector<float> *tab, aa::MelSched *sch)
{
    if (n_mels < 16 || n_mels > aa::kMelMaxMels || (n_mels & 3) || 4 * nb > 65535)
        return false;
    const int nq = n_mels / 4, pieces_max = (nb + 3) / 4;   // pieces 0 .. pieces_max - 1
    const int slots = 64 / frames_per_slot, wave_dwords = S_MAX * 64 * 4 + 2 * 64;
    if (4 * pieces_max > row_floats || pieces_max < S_MAX)
        return false;
    std::vector<int> c_lo(nq, 0), c_n(nq, 0);
    for (int q = 0; q < nq; ++q) {
        int first = -1, last = -1;
        for (int k = 0; k < nb; ++k)
            for (int i = 0; i < 4; ++i)
                if ((float)w[(size_t)(4 * q + i) * nb + k] != 0.0f) {
                    if (first < 0)
                        first = k;
                    last = k;
                }
        if (first >= 0) {
            c_lo[q] = first / 4;
            c_n[q] = last / 4 + 1 - c_lo[q];
        }
    }
    struct Place { int wave, blk0, slices, slot; };
    std::vector<Place> place(nq);
    int S = 0;
    for (int s_try = 1; s_try <= S_MAX && S == 0; ++s_try) {
        std::vector<std::pair<int, int>> items;     // (blocks, quad), largest first
        bool ok = true;
        for (int q = 0; q < nq; ++q) {
            const int sl = std::max(1, (c_n[q] + s_try - 1) / s_try);
            if (sl > 16)
                ok = false;
            items.push_back({sl, q});
        }
        if (!ok)
            continue;
        std::sort(items.begin(), items.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) {
            return a.first != b.first ? a.first > b.first : a.second < b.second;
        });
        int free_blocks[4] = {16, 16, 16, 16}, quads[4] = {0, 0, 0, 0};
        for (const auto &it : items) {
            int best = -1;
            for (int wv = 0; wv < 4; ++wv)
                if (free_blocks[wv] >= it.first && quads[wv] < slots && (best < 0 || free_blocks[wv] > free_blocks[best]))
                    best = wv;
            if (best < 0) {
                ok = false;
                break;
            }
            place[it.second] = {best, 16 - free_blocks[best], it.first, quads[best]};
            free_blocks[best] -= it.first;
            quads[best] += 1;
        }
        // every wave must emit something: its row store per chunk is counted by the chunk-top wait
        for (int wv = 0; wv < 4 && ok; ++wv)
            if (quads[wv] == 0)
                ok = false;
        if (ok)
            S = s_try;
    }
    if (S == 0)
        return false;
    *sch = aa::MelSched();
    sch->n_mels = n_mels;
    sch->n_steps = S;
    tab->assign((size_t)4 * wave_dwords, 0.0f);
    auto wave_tab = [&](int wv) { return tab->data() + (size_t)wv * wave_dwords; };
    auto words = [&](int wv) { return reinterpret_cast<uint32_t *>(wave_tab(wv) + (size_t)S_MAX * 64 * 4); };
    int hmax[4] = {0, 0, 0, 0};
    int live_steps = S;                                      // a block pushed back from the row's end multiplies later steps
    for (int q = 0; q < nq; ++q)
        hmax[place[q].wave] = std::max(hmax[place[q].wave], place[q].slices);
    for (int wv = 0; wv < 4; ++wv)
        for (int lane = 0; lane < 64; ++lane)
            words(wv)[64 + lane] = (uint32_t)hmax[wv] << 24;
    // Where a block's window of S_MAX pieces starts.  It must cover the block's pieces [p0, p1) and stay inside the row;
    // the slack left over - a narrow quad owns one or two pieces of a window of four - goes into the LDS banks: a
    // ds_read_b128 is served in four 16-lane groups (MI355X_MICROARCH.md, LDS: lanes {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31} and the same + 32), i.e. the blocks {0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15} of the B
    // operand; frame j sits at bank quarter (48 j) mod 64 and the block's piece at 4 (start mod 4) inside it, so a group
    // is conflict free iff its four starts differ mod 4.  Every permutation of residues is tried per group; blocks
    // without slack keep their start.
    static const int kB128Groups[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    struct Blk { int q = -1, h = 0, p0 = 0, p1 = 0, lo = 0, hi = 0, start = 0; };
    std::vector<Blk> blks(64);
    const int window = steps_gated ? S : S_MAX;               // pieces that multiply (2048: only the plan's n_steps)
    for (int wv = 0; wv < 4; ++wv)
        for (int b = 0; b < 16; ++b) {
            Blk &k = blks[16 * wv + b];
            k.lo = 0;
            k.hi = pieces_max - (steps_gated ? S : S_MAX);
        }
    for (int q = 0; q < nq; ++q) {
        const Place &pl = place[q];
        const int per = std::max(1, (c_n[q] + pl.slices - 1) / pl.slices);     // pieces per slice (<= S)
        for (int h = 0; h < pl.slices; ++h) {
            Blk &k = blks[16 * pl.wave + pl.blk0 + h];
            k.q = q;
            k.h = h;
            k.p0 = c_lo[q] + h * per;
            k.p1 = std::min(c_lo[q] + c_n[q], k.p0 + per);
            // the block reads S_MAX consecutive pieces from `start`.  1024 kernel: all of them multiply (unused steps carry
            // zero weights), so all of them lie inside the row; 2048 kernel (steps_gated): only the first n_steps multiply,
            // the others may run into the slack behind the row (inside the wave's scratch, never used)
            k.hi = std::max(0, std::min(k.p0, pieces_max - (steps_gated ? S : S_MAX)));
            k.lo = std::min(k.hi, std::max(0, k.p1 - window));
        }
    }
    for (int grp = 0; grp < 16; ++grp) {
        Blk *g[4];
        for (int i = 0; i < 4; ++i)
            g[i] = &blks[16 * (grp >> 2) + kB128Groups[grp & 3][i]];
        int perm[4] = {0, 1, 2, 3}, best_perm[4] = {0, 1, 2, 3}, best_hits = -1;
        do {
            int hits = 0;
            for (int i = 0; i < 4; ++i) {
                const int st = g[i]->hi - ((g[i]->hi - perm[i]) % 4 + 4) % 4;    // the largest start <= hi with the residue
                hits += st >= g[i]->lo;
            }
            if (hits > best_hits) {
                best_hits = hits;
                std::copy(perm, perm + 4, best_perm);
            }
        } while (std::next_permutation(perm, perm + 4));
        for (int i = 0; i < 4; ++i) {
            const int st = g[i]->hi - ((g[i]->hi - best_perm[i]) % 4 + 4) % 4;
            g[i]->start = st >= g[i]->lo ? st : g[i]->hi;
        }
    }
    for (int wv = 0; wv < 4; ++wv) {
        float *T = wave_tab(wv);
        uint32_t *u = words(wv);
        for (int b = 0; b < 16; ++b) {
            const Blk &k = blks[16 * wv + b];
            for (int i = 0; i < 4; ++i) {
                const int lane = 4 * b + i;             // A operand: mel 4 q + i; B operand / partial sums: frame j = i
                u[lane] = 16u * (uint32_t)k.start;
                if (k.q < 0)
                    continue;
                live_steps = std::max(live_steps, k.p1 - k.start);
                for (int piece = std::max(k.p0, k.start); piece < k.p1; ++piece)
                    for (int t = 0; t < 4; ++t)
                        T[((size_t)(piece - k.start) * 64 + lane) * 4 + t] =
                            4 * piece + t < nb ? (float)w[(size_t)(4 * k.q + i) * nb + 4 * piece + t] : 0.0f;
            }
        }
    }
    for (int q = 0; q < nq; ++q) {
        const Place &pl = place[q];
        uint32_t *u = words(pl.wave);
        for (int f = 0; f < frames_per_slot; ++f)
            u[64 + frames_per_slot * pl.slot + f] |= (uint32_t)pl.slices | ((uint32_t)pl.blk0 << 5) | ((uint32_t)q << 9);
    }
    // 1024 kernel (two frame sets of four per slot): the emission lanes (slot, frame f) read the float4 partial sums of
    // their slices, frames 4..7 from a second set that starts 64 + 4 d float4 behind the first; d = 0..3 (bits 14-15 of
    // every emission word of the wave) is the shift that leaves the fewest lanes of a ds_read_b128 group on one 16-byte
    // bank slot for THIS wave's mix of slice counts (consecutive single-slice quads: 0; two-slice quads: 1)
    if (!steps_gated && frames_per_slot == 8) {
        for (int wv = 0; wv < 4; ++wv) {
            uint32_t *u = words(wv);
            int best_d = 0, best_cost = -1;
            for (int d = 0; d < 4; ++d) {
                int cost = 0;
                for (int h = 0; h < hmax[wv]; ++h)
                    for (int grp = 0; grp < 4; ++grp) {
                        int slot_hits[16] = {0}, first[16];
                        for (int i = 0; i < 4; ++i)
                            for (int l4 = 0; l4 < 4; ++l4) {
                                const int lane = 4 * kB128Groups[grp][i] + l4;      // (lane groups = the block groups x 4)
                                const uint32_t em = u[64 + lane];
                                const int ns = (int)(em & 31u), blk0 = (int)((em >> 5) & 15u), f = lane & 7;
                                if (h >= ns)
                                    continue;                                   // (reads the zero slot: one address for all)
                                const int idx = (f >= 4 ? 64 + 4 * d : 0) + 4 * (blk0 + h) + (f & 3);
                                if (slot_hits[idx & 15] == 0 || first[idx & 15] != idx) {
                                    if (slot_hits[idx & 15] == 0)
                                        first[idx & 15] = idx;
                                    slot_hits[idx & 15] += 1;
                                }
                            }
                        int worst = 1;
                        for (int k = 0; k < 16; ++k)
                            worst = std::max(worst, slot_hits[k]);
                        cost += worst - 1;
                    }
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    best_d = d;
                }
            }
            for (int lane = 0; lane < 64; ++lane)
                u[64 + lane] |= (uint32_t)best_d << 14;
        }
    }
    S = live_steps;
    if (S > S_MAX)
        return false;
    sch->n_steps = S;
    // Verification: the kernel's dataflow on the table - per (block, mel i) the weights the MFMAs accumulate over the
    // S live steps, then the slice sums of every (quad slot, frame) lane - against the dense f32 filterbank.
    std::vector<char> emitted(n_mels, 0);
    for (int wv = 0; wv < 4; ++wv) {
        const float *T = wave_tab(wv);
        const uint32_t *u = words(wv);
        std::vector<std::vector<float>> x(64, std::vector<float>(4 * (pieces_max + S_MAX), 0.0f));   // per lane = (block, mel i)
        for (int lane = 0; lane < 64; ++lane) {
            const uint32_t off = u[lane];
            if ((off & 15u) || off / 16 + (steps_gated ? S : S_MAX) > (uint32_t)pieces_max ||
                off / 4 + 4 * S_MAX > (uint32_t)row_floats)
                return false;
            for (int s = 0; s < S_MAX; ++s)
                for (int t = 0; t < 4; ++t) {
                    const float v = T[((size_t)s * 64 + lane) * 4 + t];
                    if (v != 0.0f && s >= S)
                        return false;
                    x[lane][off / 4 + 4 * s + t] = v;
                }
        }
        for (int slot = 0; slot < slots; ++slot) {
            const uint32_t em = u[64 + frames_per_slot * slot];
            for (int f = 1; f < frames_per_slot; ++f)
                if (u[64 + frames_per_slot * slot + f] != em)
                    return false;
            if ((int)(em >> 24) != hmax[wv])
                return false;
            const int ns = (int)(em & 31u), blk0 = (int)((em >> 5) & 15u), q = (int)((em >> 9) & 31u);
            if (ns == 0)
                continue;
            if (ns > hmax[wv] || blk0 + ns > 16 || q >= nq)
                return false;
            for (int i = 0; i < 4; ++i) {
                const int mel = 4 * q + i;
                if (emitted[mel])
                    return false;
                emitted[mel] = 1;
                for (int k = 0; k < nb; ++k) {
                    float sum = 0.0f;
                    int hits = 0;
                    for (int h = 0; h < ns; ++h) {
                        const float v = x[4 * (blk0 + h) + i][k];
                        sum += v;
                        hits += v != 0.0f;
                    }
                    if (sum != (float)w[(size_t)mel * nb + k] || hits > 1)
                        return false;
                }
                for (int h = 0; h < ns; ++h)                    // nothing past the row's bins
                    for (size_t k = nb; k < x[4 * (blk0 + h) + i].size(); ++k)
                        if (x[4 * (blk0 + h) + i][k] != 0.0f)
                            return false;
            }
        }
    }
    for (int m = 0; m < n_mels; ++m)
        if (!emitted[m])
            return false;
    return true;
}

// Per-lane constants of the wave-per-frame 2048 kernel, [kWave2048Rows][64] (row meaning in
// aa_kernels.hip above stft_wave2048_kernel).
void build_wave2048_table(const float *window, std::vector<float> *tab)
{
    tab->assign((size_t)aa::kWave2048TableDwords, 0.0f);
    float *T = tab->data();
    float *W4 = T + (size_t)aa::kWave2048Rows * 64, *C4 = W4 + 8 * 64 * 4, *A4 = C4 + 8 * 64 * 4;   // streamed by the ring variant
    for (int lane = 0; lane < 64; ++lane) {
        const int b = lane >> 3;
        for (int j = 0; j < 8; ++j)
            for (int c = 0; c < 4; ++c)
                W4[(j * 64 + lane) * 4 + c] = T[(4 * j + c) * 64 + lane] = 0.5f * window[4 * (64 * j + lane) + c];
        for (int k = 1; k < 8; ++k) {
            double aA = -2.0 * kPi * (double)(lane * k) / 512.0;
            A4[((k >> 1) * 64 + lane) * 4 + 2 * (k & 1) + 0] = T[(32 + 2 * (k - 1)) * 64 + lane] = (float)std::cos(aA);
            A4[((k >> 1) * 64 + lane) * 4 + 2 * (k & 1) + 1] = T[(33 + 2 * (k - 1)) * 64 + lane] = (float)std::sin(aA);
            double aB = -2.0 * kPi * (double)(b * k) / 64.0;
            A4[((4 + (k >> 1)) * 64 + lane) * 4 + 2 * (k & 1) + 0] = T[(46 + 2 * (k - 1)) * 64 + lane] = (float)std::cos(aB);
            A4[((4 + (k >> 1)) * 64 + lane) * 4 + 2 * (k & 1) + 1] = T[(47 + 2 * (k - 1)) * 64 + lane] = (float)std::sin(aB);
        }
        for (int j = 0; j < 8; ++j) {
            const int kk = lane + 64 * j;
            double ac = -2.0 * kPi * (double)kk / 1024.0;
            C4[((j >> 1) * 64 + lane) * 4 + 2 * (j & 1) + 0] = T[(60 + 2 * j) * 64 + lane] = (float)std::cos(ac);
            C4[((j >> 1) * 64 + lane) * 4 + 2 * (j & 1) + 1] = T[(61 + 2 * j) * 64 + lane] = (float)std::sin(ac);
            double th = 2.0 * kPi * (double)kk / 2048.0;
            C4[((4 + (j >> 1)) * 64 + lane) * 4 + 2 * (j & 1) + 0] = T[(76 + 2 * j) * 64 + lane] = (float)(-std::sin(th));
            C4[((4 + (j >> 1)) * 64 + lane) * 4 + 2 * (j & 1) + 1] = T[(77 + 2 * j) * 64 + lane] = (float)(-std::cos(th));
        }
    }
}

// Frame reads of stft_wave_small_kernel without LDS bank conflicts.  A wave reads FPW frames at once
// (lane = 8 q' + gamma, gamma = (frame slot, c)); with hop a multiple of 64 floats - the reference's
// onset geometry 256 / 64, onset.rs:122-123 - every frame slot starts in the same bank, and the FPW
// frames of one ds_read_b64 collide 4-fold.  Two relabelings move a frame slot to other banks at no
// cost in the kernel: `flip` (the slot's lanes take q' ^ 4: +8 banks) and `rot` (the slot loads block
// (j + 1) mod 8 into register j: + 2 LPF banks; the radix-8 over j then yields W8^(-k) X[k], which
// the per-lane pass-A twiddles absorb).  The (flip, rot) of every slot is chosen by simulating the
// eight reads on the 64 x 4-byte banks, 32 lanes per pass.
static int small_read_passes(int n_fft, int hop, int stages, int FPW, int LPF, const int *flip, const int *rot)
{
    int total = 0;
    for (int j = 0; j < 8; ++j)
        for (int half = 0; half < 2; ++half) {
            int words[64][8], cnt[64];
            std::memset(cnt, 0, sizeof cnt);
            for (int lane = 32 * half; lane < 32 * half + 32; ++lane) {
                const int gam = lane & 7, c = gam & ((1 << stages) - 1), f = gam >> stages;
                const int crev = stages < 2 ? c : (((c & 1) << 1) | (c >> 1));
                const int q = ((lane >> 3) ^ (flip[f] ? 4 : 0)) + 8 * crev;
                const int fl = f * hop + 2 * (LPF * ((j + rot[f]) & 7) + q);    // float index of the b64
                for (int h = 0; h < 2; ++h) {
                    const int bank = (fl + h) & 63, word = fl + h;
                    bool seen = false;
                    for (int i = 0; i < cnt[bank]; ++i)
                        seen = seen || words[bank][i] == word;
                    if (!seen && cnt[bank] < 8)
                        words[bank][cnt[bank]++] = word;
                }
            }
            int worst = 1;
            for (int b = 0; b < 64; ++b)
                worst = cnt[b] > worst ? cnt[b] : worst;
            total += worst;
        }
    (void)n_fft;
    (void)FPW;
    return total;
}

// Per-lane constants of stft_wave_small_kernel (n_fft 128 / 256 / 512), [kWave256Rows][64]; the lane
// map and row meaning are described above that kernel.
void build_wave_small_table(int n_fft, int hop, const float *window, std::vector<float> *tab)
{
    tab->assign((size_t)aa::kWave256Rows * 64, 0.0f);
    float *T = tab->data();
    const int M = n_fft / 2, LPF = M / 8, FPW = 64 / LPF, stages = LPF == 8 ? 0 : LPF == 16 ? 1 : 2, PPF = M / 2 + 1;
    // (flip, rot) per frame slot: exhaustive over the 4^FPW assignments (<= 65536), fewest passes,
    // ties to the plain map
    int flip[8] = {0}, rot[8] = {0}, best = small_read_passes(n_fft, hop, stages, FPW, LPF, flip, rot);
    const bool plain = aa::experiment_knob("AA_SMALL_PLAIN_MAP", 0, 0, 1) != 0;   // experiment builds: the unrelabelled map (PMC before / after)
    if (best > 16 && !plain) {
        int bf[8] = {0}, br[8] = {0};
        for (int code = 1; code < (1 << (2 * FPW)) && best > 16; ++code) {
            int f2[8], r2[8];
            for (int f = 0; f < FPW; ++f) {
                f2[f] = (code >> (2 * f)) & 1;
                r2[f] = (code >> (2 * f + 1)) & 1;
            }
            const int passes = small_read_passes(n_fft, hop, stages, FPW, LPF, f2, r2);
            if (passes < best) {
                best = passes;
                std::memcpy(bf, f2, sizeof bf);
                std::memcpy(br, r2, sizeof br);
            }
        }
        std::memcpy(flip, bf, sizeof flip);
        std::memcpy(rot, br, sizeof rot);
    }
    for (int lane = 0; lane < 64; ++lane) {
        // before the exchange: lane = 8 q' + gamma, gamma = (frame slot, c); q = q' + 8 bitrev(c),
        // q' = (lane >> 3) ^ 4 flip[slot]
        const int gam = lane & 7, c = gam & ((1 << stages) - 1), slot = gam >> stages;
        const int qp = (lane >> 3) ^ (flip[slot] ? 4 : 0), sr = rot[slot];
        const int crev = stages < 2 ? c : (((c & 1) << 1) | (c >> 1));
        const int q = qp + 8 * crev;
        T[38 * 64 + lane] = (float)q;
        T[39 * 64 + lane] = (float)sr;
        for (int j = 0; j < 8; ++j)
            for (int cc = 0; cc < 2; ++cc)
                T[(2 * j + cc) * 64 + lane] = 0.5f * window[2 * (LPF * ((j + sr) & 7) + q) + cc];
        for (int k = 1; k < 8; ++k) {
            // W_M^(q k), times W8^(sr k): register j holds block (j + sr) mod 8
            double a = -2.0 * kPi * (double)(q * k) / (double)M - 2.0 * kPi * (double)(sr * k) / 8.0;
            T[(16 + 2 * (k - 1)) * 64 + lane] = (float)std::cos(a);
            T[(17 + 2 * (k - 1)) * 64 + lane] = (float)std::sin(a);
        }
        // DIF stage s pairs lanes that differ in lane bit s = the q bit of weight LPF >> (s + 1);
        // the odd half takes x[lower] - x[upper], times W_(LPF >> s)^(q mod (LPF >> (s + 1)))
        for (int st = 0; st < stages; ++st) {
            const int span = LPF >> st, half = span / 2;
            const bool upper = ((lane >> st) & 1) != 0;
            T[(36 + st) * 64 + lane] = upper ? -1.0f : 1.0f;
            const double a = upper ? -2.0 * kPi * (double)(q % half) / (double)span : 0.0;
            T[(30 + 2 * st) * 64 + lane] = (float)std::cos(a);
            T[(31 + 2 * st) * 64 + lane] = (float)std::sin(a);
        }
        for (int i = 0; i < 5; ++i) {
            // split twiddles -i W_N^k: 128 / 512 spread the (k, M-k) pairs of the wave's frames over the
            // lanes (pair p = lane + 64 i); 256 keeps the pairs of a frame in its own 16 lanes
            // (k = lane % 16 + 16 i, i < 4: the partner comes by DPP, see the kernel)
            const int p = lane + 64 * i;
            int kk = p < FPW * PPF ? p % PPF : 0;
            if (LPF == 16)
                kk = i < 4 ? (lane & 15) + 16 * i : 0;
            double th = 2.0 * kPi * (double)kk / (double)n_fft;
            T[(40 + 2 * i) * 64 + lane] = (float)(-std::sin(th));
            T[(41 + 2 * i) * 64 + lane] = (float)(-std::cos(th));
        }
    }
}

}  // namespace

// ---------------------------------------------------------------- plan structs --
struct aa_fft_plan {
    int len = 0;
    int device = 0;
    FftTables tabs;
    float *d_in = nullptr;   // one frame
    float *d_out = nullptr;  // one spectrum
    // the same two buffers page-locked and device-visible: the single-frame host calls run the
    // kernel directly on them (no hipMemcpy on the path: ~30 us per call became ~12)
    float *h_in = nullptr, *h_out = nullptr;
    float *h_in_dev = nullptr, *h_out_dev = nullptr;
    // batches of back-to-back frames (hop = len, no window) on the wave kernels, when len has one
    struct aa_stft_plan *batch = nullptr;
};

struct aa_stft_plan {
    aa_stft_config cfg{};
    int device = 0;
    int out_cols = 0;
    float log_floor = 1e-10f;
    std::vector<float> window_host;
    float *d_window = nullptr;
    FftTables tabs;
    // wave-per-frame paths (n_fft 1024 / 2048)
    bool wave2048 = false;
    bool wave1024 = false;
    bool wave256 = false;
    bool wave512 = false;
    bool wave128 = false;
    int frames_per_wg = 8;
    float *d_w1024_tab = nullptr;
    // mel / MFCC post-ops (build-defined)
    int n_mels = 0, n_mels_pad = 0, n_mfcc = 0;
    float *d_melT = nullptr;       // [n_bins_pad x n_mels_pad]
    int *d_tile_k = nullptr;       // [n_mels_pad/32][2]
    float *d_dct = nullptr;        // DCT-II matrix transposed, [n_mels_pad32 x n_mfcc_pad]
    int *d_dct_tile_k = nullptr;   // full K range per 32-coefficient tile
    int n_mfcc_pad = 0;
    float *d_ws_power = nullptr;   // [kMelSlab x n_bins], GEMM path only (allocated on first use)
    float *d_ws_logmel = nullptr;  // [ws_logmel_rows x n_mels]
    int64_t ws_logmel_rows = 0;
    // fused triangular mel reduction inside the wave kernels (n_fft 1024 / 2048)
    bool fused_mel = false;
    float *d_dct_lane = nullptr;   // its per-lane A-operand table [tile][8][64] float4
    float *d_mel_par = nullptr;    // [kMelParTableDwords]: the fallback gather (kLogMel)
    aa::MelSched mel_sched;
    // r4: the mel contraction on v_mfma_f32_4x4x1_16b_f32 (kLogMelM / kMfccM; n_fft 1024, 8 frames per chunk, n_mels % 4 == 0)
    bool mel_mfma = false, mfcc_mfma = false;
    float *d_mel_mfma = nullptr;   // [kMelMfmaTableDwords]
    aa::MelSched mel_mfma_sched;
    // host-execute workspace and the two-stream pipeline used for page-locked host buffers
    hipStream_t pipe_stream[2] = {nullptr, nullptr};
    hipEvent_t pipe_event = nullptr;
    float *d_ws_in = nullptr;
    size_t ws_in_floats = 0;
    float *d_ws_out = nullptr;
    size_t ws_out_floats = 0;
};

extern "C" {

const char *aa_last_error(void) { return g_err; }
int aa_abi_version(void) { return AA_ABI_VERSION; }

int aa_device_count(int *count)
{
    if (!count)
        return fail(AA_ERR_ARG, "aa_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(AA_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return AA_OK;
}

int aa_hann_window(int n, float *out_host)
{
    if (n <= 0 || !out_host)
        return fail(AA_ERR_ARG, "aa_hann_window: n=%d out=%p", n, (void *)out_host);
    hann_f32(n, out_host);
    return AA_OK;
}

// ------------------------------------------------------------------ FftProcessor --
int aa_fft_create(int len, aa_fft_plan **out_plan)
{
    if (!out_plan)
        return fail(AA_ERR_ARG, "aa_fft_create: out_plan is NULL");
    *out_plan = nullptr;
    int rc = check_len(len);
    if (rc != AA_OK)
        return rc;
    aa_fft_plan *p = new (std::nothrow) aa_fft_plan();
    if (!p)
        return fail(AA_ERR_NOMEM, "aa_fft_create: out of host memory");
    p->len = len;
    hipError_t e = hipGetDevice(&p->device);
    if (e != hipSuccess) {
        delete p;
        return fail(AA_ERR_HIP, "hipGetDevice: %s (no usable HIP device; there is no CPU fallback)",
                    hipGetErrorString(e));
    }
    rc = build_fft_tables(len, &p->tabs);
    if (rc == AA_OK) {
        hipError_t e1 = hipMalloc(&p->d_in, sizeof(float) * (size_t)(len + 2));
        hipError_t e2 = hipMalloc(&p->d_out, sizeof(float) * (size_t)(len + 2));
        if (e1 != hipSuccess || e2 != hipSuccess)
            rc = fail(AA_ERR_HIP, "hipMalloc(frame buffers): %s",
                      hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        const size_t bytes = sizeof(float) * (size_t)(len + 2);
        if (rc == AA_OK &&
            (hipHostMalloc((void **)&p->h_in, bytes, hipHostMallocDefault) != hipSuccess ||
             hipHostMalloc((void **)&p->h_out, bytes, hipHostMallocDefault) != hipSuccess ||
             hipHostGetDevicePointer((void **)&p->h_in_dev, p->h_in, 0) != hipSuccess ||
             hipHostGetDevicePointer((void **)&p->h_out_dev, p->h_out, 0) != hipSuccess)) {
            (void)hipGetLastError();          // no page-locked memory: the copying path stays
            if (p->h_in)
                (void)hipHostFree(p->h_in);
            if (p->h_out)
                (void)hipHostFree(p->h_out);
            p->h_in = p->h_out = p->h_in_dev = p->h_out_dev = nullptr;
        }
    }
    if (rc == AA_OK && (len == 128 || len == 256 || len == 512 || len == 1024 || len == 2048)) {
        aa_stft_config cfg{};
        cfg.n_fft = len;
        cfg.hop = len;
        cfg.window = AA_WINDOW_RECT;
        cfg.output = AA_OUT_COMPLEX;
        if (aa_stft_plan_create(&cfg, nullptr, &p->batch) != AA_OK)
            p->batch = nullptr;            // the generic kernel still serves the batch calls
    }
    if (rc != AA_OK) {
        aa_fft_destroy(p);
        return rc;
    }
    *out_plan = p;
    return AA_OK;
}

int aa_fft_destroy(aa_fft_plan *p)
{
    if (!p)
        return AA_OK;
    DeviceGuard g(p->device);
    free_fft_tables(&p->tabs);
    if (p->batch)
        aa_stft_plan_destroy(p->batch);
    if (p->d_in)
        (void)hipFree(p->d_in);
    if (p->d_out)
        (void)hipFree(p->d_out);
    if (p->h_in)
        (void)hipHostFree(p->h_in);
    if (p->h_out)
        (void)hipHostFree(p->h_out);
    delete p;
    return AA_OK;
}

int aa_fft_len(const aa_fft_plan *p) { return p ? p->len : 0; }

int aa_fft_forward_batch_device(aa_fft_plan *p, const float *in_dev, float *out_dev, int64_t batch,
                                void *stream)
{
    if (!p || !in_dev || !out_dev || batch < 0)
        return fail(AA_ERR_ARG, "aa_fft_forward_batch_device: bad argument");
    DeviceGuard g(p->device);
    if (!g.ok)
        return fail(AA_ERR_HIP, "cannot select device %d", p->device);
    // frames back to back = an STFT with hop = len and no window: the wave kernels take it when
    // the input is 16-byte aligned, the generic kernel otherwise
    if (p->batch && batch > 0 && (reinterpret_cast<uintptr_t>(in_dev) & 15u) == 0)
        return aa_stft_execute_device(p->batch, in_dev, batch * (int64_t)p->len, 0, batch, out_dev, stream);
    AA_HIP(aa::launch_generic_forward(p->tabs.pow2, aa::kComplex, in_dev, out_dev, nullptr,
                                      p->tabs.tw_m, p->tabs.tw_split, p->len, 0, batch, p->len,
                                      1e-10f, (hipStream_t)stream));
    return AA_OK;
}

int aa_fft_inverse_batch_device(aa_fft_plan *p, const float *spec_dev, float *out_dev,
                                int64_t batch, void *stream)
{
    if (!p || !spec_dev || !out_dev || batch < 0)
        return fail(AA_ERR_ARG, "aa_fft_inverse_batch_device: bad argument");
    DeviceGuard g(p->device);
    if (!g.ok)
        return fail(AA_ERR_HIP, "cannot select device %d", p->device);
    AA_HIP(aa::launch_generic_inverse(p->tabs.pow2, spec_dev, out_dev, p->tabs.tw_m,
                                      p->tabs.tw_split, p->len, batch, (hipStream_t)stream));
    return AA_OK;
}

int aa_fft_forward(aa_fft_plan *p, const float *in_host, float *out_host)
{
    if (!p || !in_host || !out_host)
        return fail(AA_ERR_ARG, "aa_fft_forward: NULL argument");
    DeviceGuard g(p->device);
    if (!g.ok)
        return fail(AA_ERR_HIP, "cannot select device %d", p->device);
    const size_t nin = sizeof(float) * (size_t)p->len, nout = sizeof(float) * (size_t)(p->len + 2);
    if (p->h_in_dev) {
        std::memcpy(p->h_in, in_host, nin);
        int rc = aa_fft_forward_batch_device(p, p->h_in_dev, p->h_out_dev, 1, nullptr);
        if (rc != AA_OK)
            return rc;
        AA_HIP(hipStreamSynchronize(nullptr));
        std::memcpy(out_host, p->h_out, nout);
        return AA_OK;
    }
    AA_HIP(hipMemcpyAsync(p->d_in, in_host, nin, hipMemcpyHostToDevice, nullptr));
    int rc = aa_fft_forward_batch_device(p, p->d_in, p->d_out, 1, nullptr);
    if (rc != AA_OK)
        return rc;
    AA_HIP(hipMemcpyAsync(out_host, p->d_out, nout, hipMemcpyDeviceToHost, nullptr));
    AA_HIP(hipStreamSynchronize(nullptr));
    return AA_OK;
}

int aa_fft_inverse(aa_fft_plan *p, const float *spec_host, float *out_host)
{
    if (!p || !spec_host || !out_host)
        return fail(AA_ERR_ARG, "aa_fft_inverse: NULL argument");
    DeviceGuard g(p->device);
    if (!g.ok)
        return fail(AA_ERR_HIP, "cannot select device %d", p->device);
    const size_t nspec = sizeof(float) * (size_t)(p->len + 2), nout = sizeof(float) * (size_t)p->len;
    if (p->h_in_dev) {
        std::memcpy(p->h_out, spec_host, nspec);
        int rc = aa_fft_inverse_batch_device(p, p->h_out_dev, p->h_in_dev, 1, nullptr);
        if (rc != AA_OK)
            return rc;
        AA_HIP(hipStreamSynchronize(nullptr));
        std::memcpy(out_host, p->h_in, nout);
        return AA_OK;
    }
    AA_HIP(hipMemcpyAsync(p->d_out, spec_host, nspec, hipMemcpyHostToDevice, nullptr));
    int rc = aa_fft_inverse_batch_device(p, p->d_out, p->d_in, 1, nullptr);
    if (rc != AA_OK)
        return rc;
    AA_HIP(hipMemcpyAsync(out_host, p->d_in, nout, hipMemcpyDeviceToHost, nullptr));
    AA_HIP(hipStreamSynchronize(nullptr));
    return AA_OK;
}

// ------------------------------------------------------------------ batched STFT --
int aa_stft_plan_create(const aa_stft_config *cfg, const float *custom_window_host,
                        aa_stft_plan **out_plan)
{
    if (!cfg || !out_plan)
        return fail(AA_ERR_ARG, "aa_stft_plan_create: NULL argument");
    *out_plan = nullptr;
    int rc = check_len(cfg->n_fft);
    if (rc != AA_OK)
        return rc;
    if (cfg->hop <= 0)
        return fail(AA_ERR_ARG, "hop must be positive (got %d)", cfg->hop);
    if (cfg->window < AA_WINDOW_HANN_PERIODIC || cfg->window > AA_WINDOW_CUSTOM)
        return fail(AA_ERR_ARG, "unknown window kind %d", cfg->window);
    if (cfg->window == AA_WINDOW_CUSTOM && !custom_window_host)
        return fail(AA_ERR_ARG, "AA_WINDOW_CUSTOM needs custom_window_host");
    if (cfg->output < AA_OUT_COMPLEX || cfg->output > AA_OUT_MFCC)
        return fail(AA_ERR_ARG, "unknown output kind %d", cfg->output);
    const bool mel_out = cfg->output == AA_OUT_LOGMEL || cfg->output == AA_OUT_MFCC;
    if (mel_out && (cfg->n_mels <= 0 || cfg->n_mels > 4096 || cfg->sample_rate <= 0))
        return fail(AA_ERR_ARG, "mel output needs 0 < n_mels <= 4096 and a sample rate (got %d, %d)",
                    cfg->n_mels, cfg->sample_rate);
    if (cfg->output == AA_OUT_MFCC && (cfg->n_mfcc <= 0 || cfg->n_mfcc > cfg->n_mels))
        return fail(AA_ERR_ARG, "MFCC needs 0 < n_mfcc <= n_mels (got %d)", cfg->n_mfcc);

    aa_stft_plan *p = new (std::nothrow) aa_stft_plan();
    if (!p)
        return fail(AA_ERR_NOMEM, "aa_stft_plan_create: out of host memory");
    p->cfg = *cfg;
    p->log_floor = cfg->log_floor > 0.0f ? cfg->log_floor : 1e-10f;
    const int nb = cfg->n_fft / 2 + 1;
    p->out_cols = cfg->output == AA_OUT_COMPLEX  ? 2 * nb
                  : cfg->output == AA_OUT_LOGMEL ? cfg->n_mels
                  : cfg->output == AA_OUT_MFCC   ? cfg->n_mfcc
                                                 : nb;
    hipError_t e = hipGetDevice(&p->device);
    if (e != hipSuccess) {
        delete p;
        return fail(AA_ERR_HIP, "hipGetDevice: %s (no usable HIP device; there is no CPU fallback)",
                    hipGetErrorString(e));
    }
    p->window_host.resize(cfg->n_fft);
    if (cfg->window == AA_WINDOW_HANN_PERIODIC)
        hann_f32(cfg->n_fft, p->window_host.data());
    else if (cfg->window == AA_WINDOW_RECT)
        std::fill(p->window_host.begin(), p->window_host.end(), 1.0f);
    else
        std::memcpy(p->window_host.data(), custom_window_host, sizeof(float) * cfg->n_fft);

    auto setup = [&]() -> int {
        AA_HIP(hipMalloc(&p->d_window, sizeof(float) * cfg->n_fft));
        AA_HIP(hipMemcpy(p->d_window, p->window_host.data(), sizeof(float) * cfg->n_fft,
                         hipMemcpyHostToDevice));
        int r = build_fft_tables(cfg->n_fft, &p->tabs);
        if (r != AA_OK)
            return r;
        if (mel_out) {
            const int n_mels = cfg->n_mels, n_mels_pad = (n_mels + 31) / 32 * 32;
            const int nb_pad = (nb + 31) / 32 * 32;
            if (n_mels_pad > 128)
                return fail(AA_ERR_UNSUPPORTED, "n_mels > 128 is not built (got %d)", n_mels);
            const double fmax = cfg->fmax > 0.0f ? cfg->fmax : cfg->sample_rate / 2.0;
            std::vector<double> w;
            mel_filterbank(cfg->sample_rate, cfg->n_fft, n_mels, cfg->fmin, fmax, &w);
            std::vector<float> melT((size_t)nb_pad * n_mels_pad, 0.0f);
            std::vector<int> tile_k(2 * (n_mels_pad / 32));
            for (int t = 0; t < n_mels_pad / 32; ++t) {
                int lo = nb_pad, hi = 0;
                for (int m = 32 * t; m < 32 * t + 32 && m < n_mels; ++m)
                    for (int k = 0; k < nb; ++k)
                        if (w[(size_t)m * nb + k] != 0.0) {
                            lo = std::min(lo, k);
                            hi = std::max(hi, k + 1);
                        }
                if (hi <= lo)
                    lo = hi = 0;
                tile_k[2 * t] = lo / 32 * 32;
                tile_k[2 * t + 1] = (hi + 31) / 32 * 32;
            }
            for (int m = 0; m < n_mels; ++m)
                for (int k = 0; k < nb; ++k)
                    melT[(size_t)k * n_mels_pad + m] = (float)w[(size_t)m * nb + k];
            p->n_mels = n_mels;
            p->n_mels_pad = n_mels_pad;
            p->n_mfcc = cfg->n_mfcc;
            AA_HIP(hipMalloc(&p->d_melT, melT.size() * sizeof(float)));
            AA_HIP(hipMemcpy(p->d_melT, melT.data(), melT.size() * sizeof(float), hipMemcpyHostToDevice));
            AA_HIP(hipMalloc(&p->d_tile_k, tile_k.size() * sizeof(int)));
            AA_HIP(hipMemcpy(p->d_tile_k, tile_k.data(), tile_k.size() * sizeof(int), hipMemcpyHostToDevice));
            if (cfg->output == AA_OUT_MFCC) {
                // DCT-II, orthonormal: d[c][m] = s_c cos(pi c (2m+1) / (2M)), s_0 = sqrt(1/M), else
                // sqrt(2/M); stored transposed [m][c] (B operand of the GEMM), zero padded
                const int n_mfcc_pad = (cfg->n_mfcc + 31) / 32 * 32;
                std::vector<float> dct((size_t)n_mels_pad * n_mfcc_pad, 0.0f);
                for (int c = 0; c < cfg->n_mfcc; ++c)
                    for (int m = 0; m < n_mels; ++m) {
                        double v = std::cos(kPi * c * (2.0 * m + 1.0) / (2.0 * n_mels)) * std::sqrt(2.0 / n_mels);
                        if (c == 0)
                            v *= std::sqrt(0.5);
                        dct[(size_t)m * n_mfcc_pad + c] = (float)v;
                    }
                std::vector<int> dtk(2 * (n_mfcc_pad / 32));
                for (int t = 0; t < n_mfcc_pad / 32; ++t) {
                    dtk[2 * t] = 0;
                    dtk[2 * t + 1] = n_mels_pad;
                }
                p->n_mfcc_pad = n_mfcc_pad;
                AA_HIP(hipMalloc(&p->d_dct, dct.size() * sizeof(float)));
                AA_HIP(hipMemcpy(p->d_dct, dct.data(), dct.size() * sizeof(float), hipMemcpyHostToDevice));
                AA_HIP(hipMalloc(&p->d_dct_tile_k, dtk.size() * sizeof(int)));
                AA_HIP(hipMemcpy(p->d_dct_tile_k, dtk.data(), dtk.size() * sizeof(int), hipMemcpyHostToDevice));
                AA_HIP(hipMalloc(&p->d_ws_logmel, sizeof(float) * (size_t)kMelSlab * n_mels));
                p->ws_logmel_rows = kMelSlab;
            }
        }
        if ((cfg->n_fft == 128 || cfg->n_fft == 256 || cfg->n_fft == 512) && cfg->hop % 4 == 0) {
            // chunk = 8 groups of FPW frames (two per wave): 64 / 32 / 16 frames at 128 / 256 / 512;
            // halved until the staging fits 64 KiB
            const int fpwave = 1024 / cfg->n_fft;   // frames per wave pass: 8 / 4 / 2
            int fpw = 8 * fpwave;
            while (fpw > fpwave && aa::wave_small_lds_dwords(cfg->n_fft, fpw, cfg->hop) > 16384)
                fpw /= 2;                           // 8 -> 4 -> 2 -> 1 groups (long hops: some waves idle)
            if (aa::wave_small_lds_dwords(cfg->n_fft, fpw, cfg->hop) <= 16384) {
                std::vector<float> tab;
                build_wave_small_table(cfg->n_fft, cfg->hop, p->window_host.data(), &tab);
                AA_HIP(hipMalloc(&p->d_w1024_tab, tab.size() * sizeof(float)));
                AA_HIP(hipMemcpy(p->d_w1024_tab, tab.data(), tab.size() * sizeof(float),
                                 hipMemcpyHostToDevice));
                p->frames_per_wg = fpw;
                p->wave256 = cfg->n_fft == 256;
                p->wave512 = cfg->n_fft == 512;
                p->wave128 = cfg->n_fft == 128;
            }
        }
        if (cfg->n_fft == 2048 && cfg->hop % 4 == 0) {   // 16-byte staging pieces and frame reads
            int fpw = 4;
            while (fpw > 1 && aa::wave2048_lds_dwords(fpw, cfg->hop) > 16384)
                --fpw;
            if (aa::wave2048_lds_dwords(fpw, cfg->hop) <= 16384) {
                std::vector<float> tab;
                build_wave2048_table(p->window_host.data(), &tab);
                AA_HIP(hipMalloc(&p->d_w1024_tab, tab.size() * sizeof(float)));
                AA_HIP(hipMemcpy(p->d_w1024_tab, tab.data(), tab.size() * sizeof(float),
                                 hipMemcpyHostToDevice));
                p->frames_per_wg = fpw;
                p->wave2048 = true;
            }
        }
        if (cfg->n_fft == 1024 && cfg->hop % 4 == 0) {
            // frames per workgroup chunk: the tunable default, shrunk until the LDS need fits 64 KiB.
            int fpw = 8;
            while (fpw > 1 && aa::wave1024_lds_dwords(fpw, cfg->hop) > 16384)
                --fpw;
            if (aa::wave1024_lds_dwords(fpw, cfg->hop) <= 16384) {
                std::vector<float> tab;
                build_wave1024_table(p->window_host.data(), &tab);
                AA_HIP(hipMalloc(&p->d_w1024_tab, tab.size() * sizeof(float)));
                AA_HIP(hipMemcpy(p->d_w1024_tab, tab.data(), tab.size() * sizeof(float),
                                 hipMemcpyHostToDevice));
                p->frames_per_wg = fpw;
                p->wave1024 = true;
            }
        }
        // Fused mel reduction in the wave kernels (the power rows never leave the CU); the slab + MFMA GEMM path stays for the
        // generic kernels and unaligned input.  Which path a plan takes follows from its parameters alone (the parity tests
        // reach every one of them that way: band counts that are no multiple of 4, other hops, n_mfcc > 64, n_fft 4096).
        // Experiment builds (-DAA_EXPERIMENTS=1) can force the slower ones for A/B runs: AA_MEL_GEMM=1 the unfused path,
        // AA_MEL_MFMA=0 the VALU gather instead of the contraction on the matrix cores, AA_MFCC_NO_FUSE=1 the DCT as a
        // second launch.
        auto selector = [](const char *name, int dflt) { return aa::experiment_knob(name, dflt, 0, 1); };
        if (mel_out && selector("AA_MEL_GEMM", 0) == 0 && (p->wave2048 || (p->wave1024 && p->frames_per_wg == 8))) {
            const double fmax = cfg->fmax > 0.0f ? cfg->fmax : cfg->sample_rate / 2.0;
            const bool use_mfma = selector("AA_MEL_MFMA", 1) != 0;
            std::vector<double> w;
            mel_filterbank(cfg->sample_rate, cfg->n_fft, cfg->n_mels, cfg->fmin, fmax, &w);
            // the contraction on v_mfma_f32_4x4x1_16b_f32: the 1024 kernel at 8 frames per chunk (log-mel, and MFCC with the
            // DCT-II fused behind it: n_mfcc <= 64), the 2048 kernel at the tuner hop (BASELINE configs[2])
            const bool fuse_dct = p->wave1024 && cfg->output == AA_OUT_MFCC && cfg->n_mfcc <= 16 * aa::kMfccTiles &&
                                  selector("AA_MFCC_NO_FUSE", 0) == 0;
            std::vector<float> mt;
            if (use_mfma && p->wave1024 && p->frames_per_wg == 8 && (cfg->output == AA_OUT_LOGMEL || fuse_dct) &&
                aa::wave1024_lds_dwords(p->frames_per_wg, cfg->hop, cfg->output == AA_OUT_MFCC ? 4 : 3) <= 16384)
                p->mel_mfma = build_mel_mfma(w, nb, cfg->n_mels, aa::kMelMfmaSteps, 8, aa::kMelMfmaRowB, false, &mt,
                                             &p->mel_mfma_sched);
            // (r5: also under MFCC at the tuner geometry - the log-mel slab the DCT launch reads comes from this kernel)
            if (use_mfma && p->wave2048 && cfg->hop == 512 && p->frames_per_wg == 4)
                p->mel_mfma = build_mel_mfma(w, nb, cfg->n_mels, aa::kMelMfmaSteps2048, 8, aa::kMelMfmaScratchDwords, true, &mt,
                                             &p->mel_mfma_sched);
            if (p->mel_mfma) {
                AA_HIP(hipMalloc(&p->d_mel_mfma, mt.size() * sizeof(float)));
                AA_HIP(hipMemcpy(p->d_mel_mfma, mt.data(), mt.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            // the fallback for everything else the wave kernels take (other hops, band counts that are no multiple of 4,
            // unaligned rows): the gather on the VALU with the weights rebuilt from the triangle edges
            const int lanes = p->wave2048 ? 64 : 32;   // one frame per wave pass at 2048, two at 1024
            std::vector<float> par;
            if ((p->wave2048 ? aa::wave2048_lds_dwords(p->frames_per_wg, cfg->hop, true)
                             : aa::wave1024_lds_dwords(p->frames_per_wg, cfg->hop, true)) <= 16384 &&
                build_mel_sched(cfg->sample_rate, cfg->n_fft, cfg->n_mels, cfg->fmin, fmax, lanes, &par, &p->mel_sched)) {
                AA_HIP(hipMalloc(&p->d_mel_par, par.size() * sizeof(float)));
                AA_HIP(hipMemcpy(p->d_mel_par, par.data(), par.size() * sizeof(float), hipMemcpyHostToDevice));
                p->fused_mel = true;
            }
            // MFCC: the DCT-II fused behind the contraction (the log-mel rows never leave the CU).  The kernel folds each
            // row about the middle of the mel axis (the DCT-II matrix is symmetric for even coefficients, antisymmetric
            // for odd ones), so the A-operand table of v_mfma_f32_16x16x4_f32 holds half rows: tile = (parity, half) = 16
            // coefficients c = 2 (16 half + i) + parity, lane = (i = lane % 16, K-group kq = lane / 16), entry
            // [tile][S][lane][t] = d[c][16 S + 4 kq + t], orthonormal DCT-II as above; zero outside n_mfcc / n_mels / 2.
            if (p->mel_mfma && fuse_dct) {
                std::vector<float> dl((size_t)aa::kMfccTableDwords, 0.0f);
                for (int tile = 0; tile < aa::kMfccTiles; ++tile)
                    for (int S = 0; S < 4; ++S)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int t = 0; t < 4; ++t) {
                                const int c = 2 * (16 * (tile & 1) + (lane & 15)) + (tile >> 1);
                                const int m = 16 * S + 4 * (lane >> 4) + t;
                                if (c >= cfg->n_mfcc || m >= cfg->n_mels / 2)
                                    continue;
                                double v = std::cos(kPi * c * (2.0 * m + 1.0) / (2.0 * cfg->n_mels)) *
                                           std::sqrt(2.0 / cfg->n_mels);
                                if (c == 0)
                                    v *= std::sqrt(0.5);
                                dl[(((size_t)tile * 4 + S) * 64 + lane) * 4 + t] = (float)v;
                            }
                AA_HIP(hipMalloc(&p->d_dct_lane, dl.size() * sizeof(float)));
                AA_HIP(hipMemcpy(p->d_dct_lane, dl.data(), dl.size() * sizeof(float), hipMemcpyHostToDevice));
                p->mfcc_mfma = true;
            }
        }
        return AA_OK;
    };
    rc = setup();
    if (rc != AA_OK) {
        aa_stft_plan_destroy(p);
        return rc;
    }
    *out_plan = p;
    return AA_OK;
}

int aa_stft_plan_destroy(aa_stft_plan *p)
{
    if (!p)
        return AA_OK;
    DeviceGuard g(p->device);
    free_fft_tables(&p->tabs);
    for (hipStream_t st : p->pipe_stream)
        if (st)
            (void)hipStreamDestroy(st);
    if (p->pipe_event)
        (void)hipEventDestroy(p->pipe_event);
    float *bufs[] = {p->d_window, p->d_w1024_tab, p->d_ws_in, p->d_ws_out, p->d_melT, p->d_dct, p->d_mel_par, p->d_mel_mfma, p->d_dct_lane,
                     p->d_ws_power, p->d_ws_logmel, reinterpret_cast<float *>(p->d_tile_k),
                     reinterpret_cast<float *>(p->d_dct_tile_k)};
    for (float *b : bufs)
        if (b)
            (void)hipFree(b);
    delete p;
    return AA_OK;
}

int64_t aa_stft_num_frames(const aa_stft_plan *p, int64_t n_samples)
{
    if (!p || n_samples < p->cfg.n_fft)
        return 0;
    return (n_samples - p->cfg.n_fft) / p->cfg.hop + 1;
}

int32_t aa_stft_out_cols(const aa_stft_plan *p) { return p ? p->out_cols : 0; }

const char *aa_stft_kernel_name(const aa_stft_plan *p)
{
    if (!p)
        return "";
    if (p->wave1024)
        return "wave1024";
    if (p->wave2048)
        return "wave2048";
    if (p->wave256)
        return "wave256";
    if (p->wave512)
        return "wave512";
    if (p->wave128)
        return "wave128";
    if (p->tabs.pow2 && p->cfg.n_fft >= 1024)
        return "wg_stockham";
    return p->tabs.pow2 ? "lds_stockham" : "dft";
}

const char *aa_stft_mel_path_name(const aa_stft_plan *p)
{
    if (!p || (p->cfg.output != AA_OUT_LOGMEL && p->cfg.output != AA_OUT_MFCC))
        return "";
    const bool mfcc = p->cfg.output == AA_OUT_MFCC;
    if (p->mel_mfma && (!mfcc || p->mfcc_mfma))
        return mfcc ? "mfma+dct" : "mfma";
    if (mfcc && p->mel_mfma && p->wave2048)
        return "mfma+gemm";      // the log-mel rows from the 2048 mel kernel, the DCT-II as a second launch
    if (p->fused_mel)
        return mfcc ? "gather+gemm" : "gather";
    return "gemm";
}

static int execute_device_impl(aa_stft_plan *p, const float *samples_dev, int64_t n_samples,
                               int64_t frame_begin, int64_t frame_end, float *out_dev, void *stream,
                               const aa::Channels *chn);

int aa_stft_execute_device(aa_stft_plan *p, const float *samples_dev, int64_t n_samples,
                           int64_t frame_begin, int64_t frame_end, float *out_dev, void *stream)
{
    return execute_device_impl(p, samples_dev, n_samples, frame_begin, frame_end, out_dev, stream, nullptr);
}

namespace {
struct ShardedCall {
    aa_stft_plan *plan;
    const float *local;
    int64_t local_n;
    float *rows;
    int cols;
};
int sharded_piece(void *user, int64_t fb, int64_t fe, void *stream)
{
    const ShardedCall *c = static_cast<const ShardedCall *>(user);
    return aa_stft_execute_device(c->plan, c->local, c->local_n, fb, fe, c->rows + fb * c->cols, stream);
}
}  // namespace

int aa_stft_execute_sharded(aa_stft_plan *p, aa_comm *cin, aa_comm *cout, const float *full_dev, int64_t n_samples,
                            int32_t root, float *local_dev, int64_t local_capacity, float *rows_dev, float *out_dev,
                            int32_t n_pieces, void *stream)
{
    if (!p || !cin || !cout)
        return fail(AA_ERR_ARG, "aa_stft_execute_sharded: NULL plan or communicator");
    if (aa_comm_device_index(cin) != p->device || aa_comm_device_index(cout) != p->device)
        return fail(AA_ERR_ARG, "aa_stft_execute_sharded: the plan lives on device %d, the communicators on %d / %d",
                    p->device, aa_comm_device_index(cin), aa_comm_device_index(cout));
    // the wave kernels stage 16-byte pieces of SAMPLES: from an odd sample buffer every piece would run on the generic
    // kernel of the same length instead - rows equal to rounding, but no longer bit-equal to the whole launch, which
    // callers of the sharded form rely on (the shards' rows ARE the whole's rows).  Refused rather than silently
    // different.  (Magnitude / complex rows may start anywhere: a shard's slice of a [frames][513] slab does; the fused
    // mel rows are stored as 16-byte pieces.)
    const bool mel_rows = p->cfg.output == AA_OUT_LOGMEL || p->cfg.output == AA_OUT_MFCC;
    if ((reinterpret_cast<uintptr_t>(local_dev) & 15u) || (mel_rows && (reinterpret_cast<uintptr_t>(rows_dev) & 15u)))
        return fail(AA_ERR_ARG, "aa_stft_execute_sharded: local_dev (and rows_dev of a mel plan) must be 16-byte aligned (%p, %p)",
                    (void *)local_dev, (void *)rows_dev);
    const int64_t total = aa_stft_num_frames(p, n_samples);
    int64_t f0 = 0, f1 = 0, s0 = 0, s1 = 0;
    int rc = aa_shard_frames(total, aa_comm_world(cin), aa_comm_rank(cin), &f0, &f1);
    if (rc == AA_OK)
        rc = aa_shard_samples(f0, f1, p->cfg.n_fft, p->cfg.hop, &s0, &s1);
    if (rc != AA_OK)
        return rc;
    ShardedCall call{p, local_dev, s1 - s0, rows_dev, aa_stft_out_cols(p)};
    return aa_sharded_pipeline(cin, cout, full_dev, n_samples, p->cfg.n_fft, p->cfg.hop, call.cols, root, local_dev,
                               local_capacity, rows_dev, out_dev, n_pieces, sharded_piece, &call, stream);
}

int aa_stft_execute_channels_device(aa_stft_plan *p, const float *samples_dev, int64_t n_samples,
                                    int64_t channel_stride, int32_t n_channels, float *out_dev,
                                    int64_t out_channel_stride, void *stream)
{
    if (!p)
        return fail(AA_ERR_ARG, "aa_stft_execute_channels_device: plan is NULL");
    if (n_channels < 1 || n_channels > 1024)
        return fail(AA_ERR_ARG, "n_channels %d outside 1..1024", (int)n_channels);
    const int64_t nf = aa_stft_num_frames(p, n_samples);
    if (n_channels > 1 && (channel_stride < n_samples || out_channel_stride < nf * aa_stft_out_cols(p)))
        return fail(AA_ERR_ARG, "channel strides (%lld samples, %lld floats) shorter than a channel (%lld, %lld)",
                    (long long)channel_stride, (long long)out_channel_stride, (long long)n_samples,
                    (long long)(nf * aa_stft_out_cols(p)));
    if (nf == 0)
        return AA_OK;
    if (!samples_dev || !out_dev)
        return fail(AA_ERR_ARG, "aa_stft_execute_channels_device: NULL buffer");
    // One launch when the plan's path is a single wave kernel of n_fft 1024 / 2048 (everything but the
    // slab paths of MFCC / mel) and the channels keep the 16-byte alignment of the first one
    const bool mel = p->cfg.output == AA_OUT_LOGMEL || p->cfg.output == AA_OUT_MFCC;
    const bool single_kernel = (p->wave1024 || p->wave2048) &&
                               (!mel || (p->fused_mel && p->cfg.output == AA_OUT_LOGMEL) ||
                                (p->mel_mfma && (p->cfg.output == AA_OUT_LOGMEL || p->mfcc_mfma)));
    const bool aligned = ((channel_stride | out_channel_stride) & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(samples_dev) & 15u) == 0 &&
                         (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0;
    if (n_channels > 1 && single_kernel && aligned) {
        aa::Channels ch;
        ch.n = n_channels;
        ch.in_stride = channel_stride;
        ch.out_stride = out_channel_stride;
        return execute_device_impl(p, samples_dev, n_samples, 0, nf, out_dev, stream, &ch);
    }
    for (int c = 0; c < n_channels; ++c) {
        const int rc = execute_device_impl(p, samples_dev + (int64_t)c * channel_stride, n_samples, 0, nf,
                                           out_dev + (int64_t)c * out_channel_stride, stream, nullptr);
        if (rc != AA_OK)
            return rc;
    }
    return AA_OK;
}

// chn != nullptr: only reached for plans whose path is one wave1024 / wave2048 launch (see above)
static int execute_device_impl(aa_stft_plan *p, const float *samples_dev, int64_t n_samples,
                               int64_t frame_begin, int64_t frame_end, float *out_dev, void *stream,
                               const aa::Channels *chn)
{
    if (!p)
        return fail(AA_ERR_ARG, "aa_stft_execute_device: plan is NULL");
    if (frame_begin == frame_end && frame_begin >= 0)
        return AA_OK;   // nothing to do; buffers may legitimately be empty / NULL
    if (!samples_dev || !out_dev)
        return fail(AA_ERR_ARG, "aa_stft_execute_device: NULL buffer");
    const int64_t total = aa_stft_num_frames(p, n_samples);
    if (frame_begin < 0 || frame_end < frame_begin || frame_end > total)
        return fail(AA_ERR_ARG, "frame range [%lld, %lld) outside [0, %lld)", (long long)frame_begin,
                    (long long)frame_end, (long long)total);
    const int64_t nf = frame_end - frame_begin;
    if (nf == 0)
        return AA_OK;
    if (nf > 0x7fffffffLL)
        return fail(AA_ERR_ARG, "more than 2^31-1 frames in one call");
    DeviceGuard g(p->device);
    if (!g.ok)
        return fail(AA_ERR_HIP, "cannot select device %d", p->device);
    hipStream_t st = (hipStream_t)stream;
    // The FFT stage: frames [fb, fb+n) -> rows of `dst` in output kind `kind`.
    auto fft_stage = [&](int kind, int64_t fb, int64_t n, float *dst) -> int {
        const float *first = samples_dev + fb * (int64_t)p->cfg.hop;
        if (p->wave1024 && (reinterpret_cast<uintptr_t>(first) & 15u) == 0) {
            AA_HIP(aa::launch_wave1024(kind, samples_dev, dst, p->d_w1024_tab, fb, n, p->cfg.hop,
                                       p->frames_per_wg, p->log_floor, st, nullptr, nullptr, nullptr, 0, chn));
        } else if ((p->wave128 || p->wave256 || p->wave512) && (reinterpret_cast<uintptr_t>(first) & 15u) == 0) {
            AA_HIP(aa::launch_wave_small(p->cfg.n_fft, kind, samples_dev, dst, p->d_w1024_tab, fb, n, p->cfg.hop,
                                         p->frames_per_wg, p->log_floor, st));
        } else if (p->wave2048 && (reinterpret_cast<uintptr_t>(first) & 15u) == 0) {
            AA_HIP(aa::launch_wave2048(kind, samples_dev, dst, p->d_w1024_tab, fb, n, p->cfg.hop,
                                       p->frames_per_wg, p->log_floor, st, nullptr, nullptr, chn));
        } else {
            AA_HIP(aa::launch_generic_forward(p->tabs.pow2, kind, samples_dev, dst, p->d_window,
                                              p->tabs.tw_m, p->tabs.tw_split, p->cfg.n_fft, fb, n,
                                              p->cfg.hop, p->log_floor, st));
        }
        return AA_OK;
    };
    if (p->cfg.output != AA_OUT_LOGMEL && p->cfg.output != AA_OUT_MFCC)
        return fft_stage(p->cfg.output, frame_begin, nf, out_dev);
    // mel / MFCC, fused: the wave kernel reduces each power row to n_mels log-mel values in LDS
    // (triangular gather); MFCC adds the dense DCT-II as an f32 MFMA GEMM over log-mel slabs.
    const float *first_sample = samples_dev + frame_begin * (int64_t)p->cfg.hop;
    if (p->mel_mfma && (reinterpret_cast<uintptr_t>(first_sample) & 15u) == 0) {
        // r4: the contraction on v_mfma_f32_4x4x1_16b_f32 inside the 1024 kernel (rows leave as dwordx4 pieces: 16-byte-aligned out)
        if (p->cfg.output == AA_OUT_LOGMEL && (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0 &&
            (!chn || (chn->out_stride & 3) == 0)) {
            if (p->wave2048)
                AA_HIP(aa::launch_wave2048(aa::kLogMelM, samples_dev, out_dev, p->d_w1024_tab, frame_begin, nf, p->cfg.hop,
                                           p->frames_per_wg, p->log_floor, st, p->d_mel_mfma, &p->mel_mfma_sched, chn));
            else
                AA_HIP(aa::launch_wave1024(aa::kLogMelM, samples_dev, out_dev, p->d_w1024_tab, frame_begin, nf, p->cfg.hop,
                                           p->frames_per_wg, p->log_floor, st, p->d_mel_mfma, &p->mel_mfma_sched, nullptr, 0, chn));
            return AA_OK;
        }
        if (p->cfg.output == AA_OUT_MFCC && p->mfcc_mfma) {
            AA_HIP(aa::launch_wave1024(aa::kMfccM, samples_dev, out_dev, p->d_w1024_tab, frame_begin, nf, p->cfg.hop,
                                       p->frames_per_wg, p->log_floor, st, p->d_mel_mfma, &p->mel_mfma_sched, p->d_dct_lane,
                                       p->cfg.n_mfcc, chn));
            return AA_OK;
        }
    }
    if (p->fused_mel && (reinterpret_cast<uintptr_t>(first_sample) & 15u) == 0) {
        auto mel_stage = [&](int64_t fb, int64_t n, float *dst) -> int {
            if (p->wave2048 && p->mel_mfma && !chn && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0)
                AA_HIP(aa::launch_wave2048(aa::kLogMelM, samples_dev, dst, p->d_w1024_tab, fb, n, p->cfg.hop,
                                           p->frames_per_wg, p->log_floor, st, p->d_mel_mfma, &p->mel_mfma_sched, chn));
            else if (p->wave2048)
                AA_HIP(aa::launch_wave2048(aa::kLogMel, samples_dev, dst,
                                           p->d_w1024_tab, fb, n, p->cfg.hop,
                                           p->frames_per_wg, p->log_floor, st, p->d_mel_par, &p->mel_sched, chn));
            else
                AA_HIP(aa::launch_wave1024(aa::kLogMel, samples_dev, dst,
                                           p->d_w1024_tab, fb, n, p->cfg.hop,
                                           p->frames_per_wg, p->log_floor, st, p->d_mel_par, &p->mel_sched,
                                           nullptr, 0, chn));
            return AA_OK;
        };
        if (p->cfg.output == AA_OUT_LOGMEL)
            return mel_stage(frame_begin, nf, out_dev);
        // log-mel rows are small (n_mels floats): one slab of up to 2^18 frames (<= 134 MB) so the
        // DCT GEMM runs as one large launch; the buffer grows on first use
        const int64_t slab = std::min<int64_t>(nf, kMfccSlab);
        if (slab > p->ws_logmel_rows) {
            (void)hipFree(p->d_ws_logmel);
            p->d_ws_logmel = nullptr;
            p->ws_logmel_rows = 0;
            AA_HIP(hipMalloc(&p->d_ws_logmel, sizeof(float) * (size_t)slab * p->n_mels));
            p->ws_logmel_rows = slab;
        }
        for (int64_t f = 0; f < nf; f += slab) {
            const int64_t n = std::min<int64_t>(slab, nf - f);
            // a slab start must keep the 16-byte input alignment the wave kernels stage with
            if ((reinterpret_cast<uintptr_t>(samples_dev + (frame_begin + f) * (int64_t)p->cfg.hop) & 15u) != 0)
                return fail(AA_ERR_ARG, "internal: unaligned MFCC slab");
            int rc = mel_stage(frame_begin + f, n, p->d_ws_logmel);
            if (rc != AA_OK)
                return rc;
            AA_HIP(aa::launch_rows_gemm(false, p->d_ws_logmel, p->n_mels, p->d_dct, p->d_dct_tile_k,
                                        out_dev + f * p->n_mfcc, n, p->n_mels, p->n_mfcc, p->n_mfcc_pad,
                                        0.0f, st));
        }
        return AA_OK;
    }
    // mel / MFCC, unfused: power spectrogram slab -> f32 MFMA filterbank (+log) -> optional DCT.
    // The slab (<= 34 MB) is re-read by the next kernel out of L2 / Infinity Cache.
    const int nb = p->cfg.n_fft / 2 + 1;
    if (!p->d_ws_power)
        AA_HIP(hipMalloc(&p->d_ws_power, sizeof(float) * (size_t)kMelSlab * nb));
    for (int64_t f = 0; f < nf; f += kMelSlab) {
        const int64_t n = std::min<int64_t>(kMelSlab, nf - f);
        int rc = fft_stage(aa::kPower, frame_begin + f, n, p->d_ws_power);
        if (rc != AA_OK)
            return rc;
        float *mel_dst = p->cfg.output == AA_OUT_MFCC ? p->d_ws_logmel : out_dev + f * p->n_mels;
        AA_HIP(aa::launch_rows_gemm(true, p->d_ws_power, nb, p->d_melT, p->d_tile_k, mel_dst, n, nb,
                                    p->n_mels, p->n_mels_pad, p->log_floor, st));
        if (p->cfg.output == AA_OUT_MFCC)
            AA_HIP(aa::launch_rows_gemm(false, p->d_ws_logmel, p->n_mels, p->d_dct, p->d_dct_tile_k,
                                        out_dev + f * p->n_mfcc, n, p->n_mels, p->n_mfcc, p->n_mfcc_pad,
                                        0.0f, st));
    }
    return AA_OK;
}

static bool is_pinned_host(const void *ptr)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();   // ordinary pageable memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

static int grow(float **buf, size_t *have, size_t want)
{
    if (*have >= want)
        return AA_OK;
    if (*buf)
        (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    AA_HIP(hipMalloc(buf, want * sizeof(float)));
    *have = want;
    return AA_OK;
}

int aa_stft_execute_host(aa_stft_plan *p, const float *samples_host, int64_t n_samples,
                         float *out_host)
{
    if (!p || !samples_host || !out_host || n_samples < 0)
        return fail(AA_ERR_ARG, "aa_stft_execute_host: bad argument");
    const int64_t nf = aa_stft_num_frames(p, n_samples);
    if (nf == 0)
        return AA_OK;
    DeviceGuard g(p->device);
    if (!g.ok)
        return fail(AA_ERR_HIP, "cannot select device %d", p->device);
    int rc = grow(&p->d_ws_in, &p->ws_in_floats, (size_t)n_samples);
    if (rc != AA_OK)
        return rc;
    rc = grow(&p->d_ws_out, &p->ws_out_floats, (size_t)nf * p->out_cols);
    if (rc != AA_OK)
        return rc;
    const size_t row_bytes = sizeof(float) * (size_t)p->out_cols;
    // chunks on two streams may overlap: only outputs that need no plan-owned workspace qualify
    // (the FFT outputs, and log-mel when the filterbank is fused into the FFT kernel)
    const bool fft_only = (p->cfg.output != AA_OUT_LOGMEL && p->cfg.output != AA_OUT_MFCC) ||
                          (p->cfg.output == AA_OUT_LOGMEL && p->fused_mel);
    const int64_t chunk = 16384;   // frames per pipeline stage
    if (fft_only && nf >= 2 * chunk && is_pinned_host(samples_host) && is_pinned_host(out_host)) {
        // Both buffers are page-locked (aa_host_register / hipHostMalloc): run H2D, kernel and
        // D2H of consecutive frame chunks on two streams so the copy engines overlap compute.
        for (int i = 0; i < 2; ++i)
            if (!p->pipe_stream[i])
                AA_HIP(hipStreamCreateWithFlags(&p->pipe_stream[i], hipStreamNonBlocking));
        if (!p->pipe_event)
            AA_HIP(hipEventCreateWithFlags(&p->pipe_event, hipEventDisableTiming));
        int64_t copied = 0;   // samples [0, copied) are on the device or in flight
        int k = 0;
        for (int64_t f0 = 0; f0 < nf; f0 += chunk, ++k) {
            const int64_t f1 = std::min(nf, f0 + chunk);
            const int64_t need = (f1 - 1) * (int64_t)p->cfg.hop + p->cfg.n_fft;
            hipStream_t st = p->pipe_stream[k & 1];
            if (k > 0)   // the halo of this chunk was copied by the previous chunk's stream
                AA_HIP(hipStreamWaitEvent(st, p->pipe_event, 0));
            AA_HIP(hipMemcpyAsync(p->d_ws_in + copied, samples_host + copied,
                                  sizeof(float) * (size_t)(need - copied), hipMemcpyHostToDevice, st));
            copied = need;
            AA_HIP(hipEventRecord(p->pipe_event, st));
            rc = aa_stft_execute_device(p, p->d_ws_in, n_samples, f0, f1, p->d_ws_out + f0 * p->out_cols, st);
            if (rc != AA_OK)
                return rc;
            AA_HIP(hipMemcpyAsync(out_host + f0 * p->out_cols, p->d_ws_out + f0 * p->out_cols,
                                  row_bytes * (size_t)(f1 - f0), hipMemcpyDeviceToHost, st));
        }
        AA_HIP(hipStreamSynchronize(p->pipe_stream[0]));
        AA_HIP(hipStreamSynchronize(p->pipe_stream[1]));
        return AA_OK;
    }
    AA_HIP(hipMemcpyAsync(p->d_ws_in, samples_host, sizeof(float) * (size_t)n_samples,
                          hipMemcpyHostToDevice, nullptr));
    rc = aa_stft_execute_device(p, p->d_ws_in, n_samples, 0, nf, p->d_ws_out, nullptr);
    if (rc != AA_OK)
        return rc;
    AA_HIP(hipMemcpyAsync(out_host, p->d_ws_out, row_bytes * (size_t)nf, hipMemcpyDeviceToHost, nullptr));
    AA_HIP(hipStreamSynchronize(nullptr));
    return AA_OK;
}

// Page-lock a caller buffer so aa_stft_execute_host can stream it at PCIe rate.
int aa_host_register(void *ptr, size_t bytes)
{
    if (!ptr || bytes == 0)
        return fail(AA_ERR_ARG, "aa_host_register: bad argument");
    AA_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return AA_OK;
}

int aa_host_unregister(void *ptr)
{
    if (!ptr)
        return fail(AA_ERR_ARG, "aa_host_unregister: NULL");
    AA_HIP(hipHostUnregister(ptr));
    return AA_OK;
}

// ----------------------------------------------------- rows placed by measurement --
// process_forward's own rows (AA_OUT_COMPLEX, fft.rs:33-35) leave the 1024 / 256 launch at ~6.4 TB/s into some output
// buffers and ~5.1 TB/s into others - a property of the PHYSICAL memory behind the buffer, for the buffer's lifetime.
// r5 (profiles/r05_rows_scan_complex.txt, r05_rows_contrast.txt): of 465 slabs of 462 MB that tile 200 GiB of one
// device, 22 are fast (95 us), in runs of neighbouring slabs; into a fast and a slow slab of one process the launch
// issues the same 7.49 M write requests of the same sizes, and in the slow one the L2 waits 1.44 x as long for DRAM
// write credits and a request stays outstanding 9 % longer - the memory side drains these 1024 concurrent row streams
// less efficiently from some physical regions than from others.  Nothing a kernel or a virtual address can steer; so
// the library does what it can: take slab after slab (all held, so each is other physical memory), time the plan's own
// launch into each, stop at the first that is clearly faster than the others, free the rest.
namespace {
float time_launches(aa_stft_plan *p, const float *samples, int64_t n_samples, int64_t n_frames, float *rows, int warm,
                    int timed, hipEvent_t e0, hipEvent_t e1, int &rc)
{
    for (int i = 0; i < warm + timed && rc == AA_OK; ++i) {
        if (i == warm && hipEventRecord(e0, nullptr) != hipSuccess)
            rc = fail(AA_ERR_HIP, "aa_rows_alloc: hipEventRecord");
        if (rc == AA_OK)
            rc = aa_stft_execute_device(p, samples, n_samples, 0, n_frames, rows, nullptr);
    }
    float ms = 0.0f;
    if (rc == AA_OK && (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                        hipEventElapsedTime(&ms, e0, e1) != hipSuccess))
        rc = fail(AA_ERR_HIP, "aa_rows_alloc: timing a candidate failed");
    return ms * 1e3f / (float)timed;
}
}  // namespace

int aa_rows_alloc(aa_stft_plan *p, int64_t n_frames, int32_t n_candidates, float **out_rows_dev, float *kept_us,
                  float *typical_us)
{
    if (!p || !out_rows_dev || n_frames < 0 || n_candidates < 1 || n_candidates > 4096)
        return fail(AA_ERR_ARG, "aa_rows_alloc: bad argument (1..4096 candidates)");
    *out_rows_dev = nullptr;
    if (kept_us)
        *kept_us = 0.0f;
    if (typical_us)
        *typical_us = 0.0f;
    aa::DeviceGuard guard(p->device);
    if (!guard.ok)
        return fail(AA_ERR_HIP, "aa_rows_alloc: cannot select device %d", p->device);
    const size_t bytes = sizeof(float) * (size_t)(n_frames > 0 ? n_frames : 1) * (size_t)p->out_cols;
    if (n_candidates == 1 || n_frames == 0) {
        AA_HIP(hipMalloc(reinterpret_cast<void **>(out_rows_dev), bytes));
        return AA_OK;
    }
    const int64_t n_samples = (n_frames - 1) * (int64_t)p->cfg.hop + p->cfg.n_fft;
    std::vector<float *> cand;
    std::vector<float> us;
    float *samples = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipMalloc(reinterpret_cast<void **>(&samples), sizeof(float) * (size_t)n_samples);
    // (any finite samples will do: the launch has no data-dependent path; 0x3C3C3C3C is 0.0115)
    if (err == hipSuccess)
        err = hipMemset(samples, 0x3C, sizeof(float) * (size_t)n_samples);
    if (err == hipSuccess)
        err = hipEventCreate(&e0);
    if (err == hipSuccess)
        err = hipEventCreate(&e1);
    int rc = err == hipSuccess ? AA_OK : fail(AA_ERR_HIP, "aa_rows_alloc: %s", hipGetErrorString(err));
    auto median = [&]() {
        std::vector<float> v(us);
        std::nth_element(v.begin(), v.begin() + (long)(v.size() / 2), v.end());
        return v[v.size() / 2];
    };
    int kept = -1;
    while (rc == AA_OK && (int)cand.size() < n_candidates) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + (size_t(2) << 30))
            break;                                   // (leave the device two GiB: choose among what there is)
        float *slab = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&slab), bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        cand.push_back(slab);
        if (cand.size() == 1)                        // the clocks up before anything is compared
            (void)time_launches(p, samples, n_samples, n_frames, slab, 0, 60, e0, e1, rc);
        us.push_back(time_launches(p, samples, n_samples, n_frames, slab, 2, 6, e0, e1, rc));
        if (rc != AA_OK || us.size() < 6)
            continue;
        const float med = median();
        if (us.back() < 0.90f * med) {
            // confirm against the slab nearest the median before the search ends on it
            size_t ref = 0;
            for (size_t i = 0; i + 1 < us.size(); ++i)
                ref = std::fabs(us[i] - med) < std::fabs(us[ref] - med) ? i : ref;
            const float a = time_launches(p, samples, n_samples, n_frames, slab, 2, 16, e0, e1, rc);
            const float b = time_launches(p, samples, n_samples, n_frames, cand[ref], 2, 16, e0, e1, rc);
            us.back() = a;
            if (rc == AA_OK && a < 0.92f * b) {
                kept = (int)cand.size() - 1;
                break;
            }
        }
    }
    if (rc == AA_OK && cand.empty())
        rc = fail(AA_ERR_NOMEM, "aa_rows_alloc: no candidate slab could be allocated");
    if (rc == AA_OK) {
        if (kept < 0) {                              // no region stands out: the fastest of what was seen
            kept = 0;
            for (size_t c = 1; c < us.size(); ++c)
                kept = us[c] < us[(size_t)kept] ? (int)c : kept;
        }
        *out_rows_dev = cand[(size_t)kept];
        if (kept_us)
            *kept_us = us[(size_t)kept];
        if (typical_us)
            *typical_us = median();
    }
    for (size_t c = 0; c < cand.size(); ++c)
        if (rc != AA_OK || (int)c != kept)
            (void)hipFree(cand[c]);
    if (samples)
        (void)hipFree(samples);
    if (e0)
        (void)hipEventDestroy(e0);
    if (e1)
        (void)hipEventDestroy(e1);
    return rc;
}

int aa_rows_free(float *rows_dev)
{
    if (rows_dev)
        AA_HIP(hipFree(rows_dev));
    return AA_OK;
}

// ----------------------------------------------------------------------- downmix --
int aa_downmix_device(const float *interleaved_dev, int64_t n_frames, int32_t channels,
                      float *mono_dev, void *stream)
{
    if (!interleaved_dev || !mono_dev || n_frames < 0 || channels <= 0)
        return fail(AA_ERR_ARG, "aa_downmix_device: bad argument");
    AA_HIP(aa::launch_downmix(interleaved_dev, mono_dev, n_frames, channels, (hipStream_t)stream));
    return AA_OK;
}

// ------------------------------------------------------------------- onset front --
int aa_onset_flux_device(const float *mags_dev, int64_t n_frames, int32_t n_bins,
                         const float *prev_row_dev, float *flux_dev, float *energy_dev, void *stream)
{
    if (!mags_dev || !flux_dev || !energy_dev || n_frames < 0 || n_bins < 2)
        return fail(AA_ERR_ARG, "aa_onset_flux_device: bad argument");
    if (n_bins > 4160)
        return fail(AA_ERR_UNSUPPORTED, "aa_onset_flux_device: n_bins > 4160 is not built (got %d)", n_bins);
    AA_HIP(aa::launch_onset_flux(mags_dev, prev_row_dev, flux_dev, energy_dev, n_frames, n_bins,
                                 (hipStream_t)stream));
    return AA_OK;
}

int aa_onset_gates_device(const float *mags_dev, int64_t n_frames, int32_t n_bins, const float *flux_dev,
                          const float *energy_dev, const float *global_floor_dev, const uint8_t *suppress_dev,
                          int32_t calibrated, float *state_dev, int32_t initialized, float *out_flux_dev,
                          float *out_ema_dev, float *out_excess_dev, int32_t *out_count_dev, uint8_t *out_flags_dev,
                          float *out_velocity_dev, void *stream)
{
    if (!mags_dev || !flux_dev || !energy_dev || !global_floor_dev || !state_dev || !out_flux_dev ||
        !out_ema_dev || !out_excess_dev || !out_count_dev || !out_flags_dev || !out_velocity_dev ||
        n_frames < 0 || n_bins < 1)
        return fail(AA_ERR_ARG, "aa_onset_gates_device: bad argument");
    if (n_bins > 4160)
        return fail(AA_ERR_UNSUPPORTED, "aa_onset_gates_device: n_bins > 4160 is not built (got %d)", n_bins);
    AA_HIP(aa::launch_onset_gates(mags_dev, flux_dev, energy_dev, global_floor_dev, state_dev, initialized,
                                  n_frames, n_bins, out_flux_dev, out_ema_dev, out_excess_dev, out_count_dev,
                                  out_flags_dev, out_velocity_dev, (hipStream_t)stream, nullptr, nullptr, 1,
                                  suppress_dev, calibrated));
    return AA_OK;
}

// ------------------------------------------------------------------- diagnostics --
int aa_diag_flux_div3_device(uint32_t first_bits, uint64_t count, uint64_t *mismatches_host)
{
    if (!mismatches_host || count > (1ull << 32))
        return fail(AA_ERR_ARG, "aa_diag_flux_div3_device: bad argument");
    unsigned long long *d = nullptr;
    AA_HIP(hipMalloc(&d, sizeof *d));
    hipError_t e = hipMemset(d, 0, sizeof *d);
    if (e == hipSuccess)
        e = aa::launch_flux_div3_check(first_bits, count, d, nullptr);
    unsigned long long h = 0;
    if (e == hipSuccess)
        e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess)
        return fail(AA_ERR_HIP, "aa_diag_flux_div3_device: %s", hipGetErrorString(e));
    *mismatches_host = h;
    return AA_OK;
}

int aa_diag_stream_copy_device(const float *src_dev, float *dst_dev, int64_t n_floats, int32_t writes_per_read,
                               void *stream)
{
    if (!src_dev || !dst_dev || n_floats < 0 || (n_floats & 3) ||
        (writes_per_read != 1 && writes_per_read != 2 && writes_per_read != 4) ||
        ((reinterpret_cast<uintptr_t>(src_dev) | reinterpret_cast<uintptr_t>(dst_dev)) & 15u))
        return fail(AA_ERR_ARG, "aa_diag_stream_copy_device: n_floats %% 4 == 0, 16-byte aligned buffers, 1, 2 or 4 writes per read");
    AA_HIP(aa::launch_stream_copy(src_dev, dst_dev, n_floats, writes_per_read, (hipStream_t)stream));
    return AA_OK;
}

// ------------------------------------------------------------------- tuner front --
int aa_tuner_floor_device(const float *mags_dev, int64_t n_frames, int32_t n_bins,
                          const float *global_floor_dev, float *state_dev, int32_t initialized,
                          float *eff_floor_dev, void *stream)
{
    if (!mags_dev || !global_floor_dev || !state_dev || !eff_floor_dev || n_frames < 0 || n_bins < 1)
        return fail(AA_ERR_ARG, "aa_tuner_floor_device: bad argument");
    AA_HIP(aa::launch_tuner_floor(mags_dev, global_floor_dev, state_dev, initialized, eff_floor_dev,
                                  n_frames, n_bins, (hipStream_t)stream));
    return AA_OK;
}

int aa_tuner_pitches_device(const float *mags_dev, const float *eff_floor_dev, int64_t n_frames,
                            int32_t n_bins, float bin_width, float min_freq, float max_freq,
                            float *out_freq_dev, float *out_score_dev, int32_t *out_count_dev, void *stream)
{
    if (!mags_dev || !eff_floor_dev || !out_freq_dev || !out_score_dev || !out_count_dev || n_frames < 0 ||
        n_bins < 3 || !(bin_width > 0.0f))
        return fail(AA_ERR_ARG, "aa_tuner_pitches_device: bad argument");
    if (n_bins > 1537)   // two rows + six per-peak arrays of n_bins floats in 64 KiB of LDS
        return fail(AA_ERR_UNSUPPORTED, "aa_tuner_pitches_device: n_bins > 1537 is not built (got %d)", n_bins);
    AA_HIP(aa::launch_tuner_pitches(mags_dev, eff_floor_dev, n_frames, n_bins, bin_width, min_freq, max_freq,
                                    out_freq_dev, out_score_dev, out_count_dev, (hipStream_t)stream));
    return AA_OK;
}

// ---- batches of independent streams (the recurrences above parallelise across streams) ----
int aa_onset_flux_batch_device(const float *mags_dev, const int64_t *frame_offsets_dev, int32_t n_streams,
                               int64_t n_frames_total, int32_t n_bins, const float *prev_rows_dev,
                               float *flux_dev, float *energy_dev, void *stream)
{
    if (!mags_dev || !frame_offsets_dev || !flux_dev || !energy_dev || n_streams < 1 || n_frames_total < 0 ||
        n_bins < 2)
        return fail(AA_ERR_ARG, "aa_onset_flux_batch_device: bad argument");
    if (n_bins > 4160)
        return fail(AA_ERR_UNSUPPORTED, "aa_onset_flux_batch_device: n_bins > 4160 is not built (got %d)", n_bins);
    AA_HIP(aa::launch_onset_flux(mags_dev, prev_rows_dev, flux_dev, energy_dev, n_frames_total, n_bins,
                                 (hipStream_t)stream, frame_offsets_dev, n_streams));
    return AA_OK;
}

int aa_onset_gates_batch_device(const float *mags_dev, const int64_t *frame_offsets_dev, int32_t n_streams,
                                int64_t n_frames_total, int32_t n_bins, const float *flux_dev,
                                const float *energy_dev, const float *global_floor_dev, const uint8_t *suppress_dev,
                                int32_t calibrated, float *state_dev, const int32_t *initialized_dev,
                                float *out_flux_dev, float *out_ema_dev, float *out_excess_dev,
                                int32_t *out_count_dev, uint8_t *out_flags_dev, float *out_velocity_dev,
                                void *stream)
{
    if (!mags_dev || !frame_offsets_dev || !flux_dev || !energy_dev || !global_floor_dev || !state_dev ||
        !out_flux_dev || !out_ema_dev || !out_excess_dev || !out_count_dev || !out_flags_dev ||
        !out_velocity_dev || n_streams < 1 || n_frames_total < 0 || n_bins < 1)
        return fail(AA_ERR_ARG, "aa_onset_gates_batch_device: bad argument");
    if (n_bins > 4160)
        return fail(AA_ERR_UNSUPPORTED, "aa_onset_gates_batch_device: n_bins > 4160 is not built (got %d)", n_bins);
    AA_HIP(aa::launch_onset_gates(mags_dev, flux_dev, energy_dev, global_floor_dev, state_dev, 0, n_frames_total,
                                  n_bins, out_flux_dev, out_ema_dev, out_excess_dev, out_count_dev, out_flags_dev,
                                  out_velocity_dev, (hipStream_t)stream, frame_offsets_dev, initialized_dev,
                                  n_streams, suppress_dev, calibrated));
    return AA_OK;
}

int aa_tuner_floor_batch_device(const float *mags_dev, const int64_t *frame_offsets_dev, int32_t n_streams,
                                int64_t n_frames_total, int32_t n_bins, const float *global_floor_dev,
                                float *state_dev, const int32_t *initialized_dev, float *eff_floor_dev,
                                void *stream)
{
    if (!mags_dev || !frame_offsets_dev || !global_floor_dev || !state_dev || !eff_floor_dev || n_streams < 1 ||
        n_streams > 65535 || n_frames_total < 0 || n_bins < 1)
        return fail(AA_ERR_ARG, "aa_tuner_floor_batch_device: bad argument");
    AA_HIP(aa::launch_tuner_floor(mags_dev, global_floor_dev, state_dev, 0, eff_floor_dev, n_frames_total, n_bins,
                                  (hipStream_t)stream, frame_offsets_dev, initialized_dev, n_streams));
    return AA_OK;
}

// ---- input conditioning of the reducer thread (mod.rs:351-478, dynamics.rs:152-360) ----
// Sample-sequential IIRs and a per-slot tracker with sorts: host work in the reference and here
// (one stream has no parallelism to offer a GPU); it supplies what the detectors need from outside
// the path - conditioned samples and noise_floor_db per slot.
#pragma clang fp contract(off)
namespace {
struct Biquad {
    float b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0, x1 = 0, x2 = 0, y1 = 0, y2 = 0;
    void design(float sample_rate, float freq, bool lowpass)
    {
        const float w0 = 2.0f * 3.14159265358979323846f * freq / sample_rate;
        const float cw = std::cos(w0), sw = std::sin(w0);
        const float alpha = sw / (2.0f * 0.707f);
        const float nb0 = lowpass ? (1.0f - cw) / 2.0f : (1.0f + cw) / 2.0f;
        const float nb1 = lowpass ? 1.0f - cw : -(1.0f + cw);
        const float a0 = 1.0f + alpha;
        b0 = nb0 / a0;
        b1 = nb1 / a0;
        b2 = nb0 / a0;
        a1 = (-2.0f * cw) / a0;
        a2 = (1.0f - alpha) / a0;
    }
    float step(float x)
    {
        const float y = b0 * x + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
        x2 = x1;
        x1 = x;
        y2 = y1;
        y1 = y;
        return y;
    }
};
float to_db(float linear) { return 20.0f * std::log10(std::max(linear, 1e-9f)); }
}  // namespace

// A ring of the last N slot levels together with the same values in ascending order.  The
// reference copies and sorts the window every slot (dynamics.rs:211-217, 291-301: 5000 values, ~0.1 ms
// - nothing at 47 slots a second, 4.5 of the 5.2 s an offline pass over ten minutes took here); an
// order statistic depends on the multiset only, so the sorted copy is kept current by one erase and
// one insert instead: the same percentiles, bit for bit.
struct SortedRing {
    std::vector<float> ring, sorted;
    size_t pos = 0;
    bool full = false;
    explicit SortedRing(size_t n) : ring(n, 0.0f) { sorted.reserve(n); }
    size_t count() const { return sorted.size(); }
    void push(float v)
    {
        if (full) {
            auto old = std::lower_bound(sorted.begin(), sorted.end(), ring[pos]);
            if (old != sorted.end())
                sorted.erase(old);
        }
        ring[pos] = v;
        sorted.insert(std::upper_bound(sorted.begin(), sorted.end(), v), v);
        pos = (pos + 1) % ring.size();
        full = full || pos == 0;
    }
};

struct aa_conditioner {
    int slot_len = 1024;
    Biquad hp, lp;
    float gate_threshold = 0, envelope = 0, release = 0;
    int64_t hold_samples = 0, hold_left = 0;
    // DynamicsTracker: 256 quiet slots (~12 s) for the noise floor, 5000 playing slots (~8 min) for the
    // session median and the AGC target
    SortedRing quiet{256}, playing{5000};
    float gain = 1.0f, target_db = -18.0f, max_boost_db = 100.0f, smooth_alpha = 0, silence_alpha = 0;
    float active_snr_db = 20.0f, bootstrap_floor_db = -55.0f;
};

int aa_conditioner_create(float sample_rate, int32_t slot_len, aa_conditioner **out)
{
    if (!out || !(sample_rate > 0.0f) || slot_len < 1)
        return fail(AA_ERR_ARG, "aa_conditioner_create: bad argument");
    aa_conditioner *c = new (std::nothrow) aa_conditioner();
    if (!c)
        return fail(AA_ERR_NOMEM, "aa_conditioner_create: out of memory");
    c->slot_len = slot_len;
    c->hp.design(sample_rate, 40.0f, false);
    c->lp.design(sample_rate, 14000.0f, true);
    c->gate_threshold = std::pow(10.0f, -60.0f / 20.0f);
    c->release = std::exp(-1.0f / (0.040f * sample_rate));
    c->hold_samples = (int64_t)(0.020f * sample_rate);
    const float slot_rate = sample_rate / (float)slot_len;
    c->smooth_alpha = 1.0f - std::exp(-1.0f / (240.0f * slot_rate));
    c->silence_alpha = 1.0f - std::exp(-1.0f / (10.0f * slot_rate));
    *out = c;
    return AA_OK;
}

int aa_conditioner_destroy(aa_conditioner *c)
{
    delete c;
    return AA_OK;
}

int aa_conditioner_process(aa_conditioner *c, float *samples_host, int64_t n_slots, float *noise_floor_db,
                           float *rms_db_out, float *gain_db, float *session_median_db, int32_t *level)
{
    if (!c || !samples_host || n_slots < 0 || !noise_floor_db || !rms_db_out || !gain_db || !session_median_db ||
        !level)
        return fail(AA_ERR_ARG, "aa_conditioner_process: bad argument");
    const int n = c->slot_len;
    for (int64_t s = 0; s < n_slots; ++s) {
        float *slot = samples_host + s * (int64_t)n;
        // 40 Hz high-pass, 14 kHz low-pass, envelope gate (instant attack, 40 ms release, 20 ms hold)
        for (int i = 0; i < n; ++i) {
            const float x = c->lp.step(c->hp.step(slot[i]));
            const float mag = std::fabs(x);
            if (mag > c->envelope) {
                c->envelope = mag;
                c->hold_left = c->hold_samples;
            } else {
                c->envelope = c->release * c->envelope + (1.0f - c->release) * mag;
            }
            float g = 1.0f;
            if (c->envelope < c->gate_threshold) {
                if (c->hold_left > 0) {
                    c->hold_left -= 1;
                } else {
                    const float ratio = c->envelope / c->gate_threshold;
                    g = ratio * ratio * ratio * ratio;
                }
            }
            slot[i] = x * g;
        }
        // slot loudness and the noise floor = 10th percentile of the quiet slots' RMS
        float sum_sq = 0.0f;
        for (int i = 0; i < n; ++i)
            sum_sq += slot[i] * slot[i];
        const float rms = std::sqrt(sum_sq / (float)n);
        const float rms_db = to_db(rms);
        // (an empty history reads as its first, still zero, element: dynamics.rs:204-217)
        const size_t quiet_n = std::max<size_t>(c->quiet.count(), 1);
        const float p10 = c->quiet.count() ? c->quiet.sorted[(size_t)((float)(quiet_n - 1) * 0.10f)] : 0.0f;
        const float floor_now_db = to_db(std::max(p10, 1e-9f));
        const float gate_floor_db = quiet_n >= 32 ? floor_now_db : c->bootstrap_floor_db;
        const bool active = rms_db > gate_floor_db + c->active_snr_db;
        bool broadband = false;
        if (active) {   // Gaussian-like kurtosis at a low level = ambient noise, not playing
            const float mean_sq = rms * rms;
            float quad = 0.0f;
            for (int i = 0; i < n; ++i) {
                const float s2 = slot[i] * slot[i];
                quad += s2 * s2;
            }
            const float mean_quad = quad / (float)n;
            const float kurt = mean_sq > 1e-18f ? mean_quad / (mean_sq * mean_sq) : 3.0f;
            broadband = kurt >= 2.75f && kurt <= 3.8f && rms_db < -45.0f;
        }
        const bool playing = active && !broadband;
        if (!active || broadband)
            c->quiet.push(rms);
        if (playing)
            c->playing.push(rms);
        // session statistics: median and 95th percentile of the playing slots
        const size_t play_n = c->playing.count();
        float want_gain_db = 0.0f, median_db = rms_db;
        if (play_n > 0) {
            const std::vector<float> &sorted = c->playing.sorted;
            median_db = to_db(std::max(sorted[(play_n - 1) / 2], 1e-9f));
            const float p95_db = to_db(std::max(sorted[(size_t)((float)(play_n - 1) * 0.95f)], 1e-9f));
            want_gain_db = std::min(std::max(c->target_db - p95_db, 0.0f), c->max_boost_db);
        }
        if (playing)
            c->gain += c->smooth_alpha * (std::pow(10.0f, want_gain_db / 20.0f) - c->gain);
        else
            c->gain += c->silence_alpha * (1.0f - c->gain);
        float peak = 0.0f;
        for (int i = 0; i < n; ++i)
            peak = std::max(peak, std::fabs(slot[i]));
        const float applied = std::min(c->gain, 0.97f / std::max(peak, 1e-9f));
        for (int i = 0; i < n; ++i)
            slot[i] *= applied;
        int32_t lv = 0;   // Silence, then Ppp .. Fff by level relative to the session median
        if (playing) {
            const float rel = rms_db - median_db;
            static const float edges[7] = {-15.0f, -9.0f, -4.5f, -1.5f, 1.5f, 4.5f, 9.0f};
            lv = 1;
            for (float e : edges)
                lv += rel < e ? 0 : 1;
        }
        noise_floor_db[s] = floor_now_db;
        rms_db_out[s] = rms_db;
        gain_db[s] = to_db(applied);
        session_median_db[s] = median_db;
        level[s] = lv;
    }
    return AA_OK;
}

// PitchTracker (src/audio_io/stft.rs:28-117): sequential hit / miss hysteresis over <= 8 raw
// pitches per frame - host logic, like in the reference (nothing here is worth a kernel).
struct aa_pitch_tracker {
    struct Track {
        float freq, score;
        int life;
    };
    std::vector<Track> tracks;
};

int aa_pitch_tracker_create(aa_pitch_tracker **out)
{
    if (!out)
        return fail(AA_ERR_ARG, "aa_pitch_tracker_create: out is NULL");
    *out = new (std::nothrow) aa_pitch_tracker();
    return *out ? AA_OK : fail(AA_ERR_NOMEM, "aa_pitch_tracker_create: out of memory");
}

int aa_pitch_tracker_destroy(aa_pitch_tracker *t)
{
    delete t;
    return AA_OK;
}

int aa_pitch_tracker_process(aa_pitch_tracker *t, const float *freq_host, const float *score_host,
                             const int32_t *count_host, const uint8_t *onset_host, int64_t n_frames,
                             float *out_freq_host, float *out_score_host, int32_t *out_count_host)
{
    if (!t || !freq_host || !score_host || !count_host || !out_freq_host || !out_score_host || !out_count_host ||
        n_frames < 0)
        return fail(AA_ERR_ARG, "aa_pitch_tracker_process: bad argument");
    constexpr int kDisplayThreshold = 2, kMaxLife = 3, kCap = AA_PITCH_TRACKS_MAX;
    constexpr float kTolerance = 0.03f;
    std::vector<char> matched;
    for (int64_t f = 0; f < n_frames; ++f) {
        const bool onset = onset_host && onset_host[f] != 0;
        auto &tr = t->tracks;
        matched.assign(tr.size(), 0);
        const int n_raw = std::min<int32_t>(std::max<int32_t>(count_host[f], 0), 8);
        for (int r = 0; r < n_raw; ++r) {
            const float raw_freq = freq_host[f * 8 + r], raw_score = score_host[f * 8 + r];
            bool found = false;
            for (size_t i = 0; i < tr.size() && !found; ++i) {
                if (matched[i] || !(std::fabs(tr[i].freq - raw_freq) / tr[i].freq < kTolerance))
                    continue;
                // an onset snaps to the new frequency, otherwise the track is EMA-blended
                tr[i].freq = onset ? raw_freq : tr[i].freq * 0.6f + raw_freq * 0.4f;
                tr[i].score = raw_score;
                tr[i].life = std::min(tr[i].life + 1, kMaxLife);
                matched[i] = 1;
                found = true;
            }
            if (!found) {
                tr.push_back({raw_freq, raw_score, 1});
                matched.push_back(1);
            }
        }
        // misses decay (or die at once on an onset), dead tracks are reaped, stable ones emitted
        int n_out = 0;
        size_t w = 0;
        for (size_t i = 0; i < tr.size(); ++i) {
            if (!matched[i])
                tr[i].life = onset ? 0 : tr[i].life - 1;
            if (tr[i].life <= 0)
                continue;
            if (tr[i].life >= kDisplayThreshold && n_out < kCap) {
                out_freq_host[f * kCap + n_out] = tr[i].freq;
                out_score_host[f * kCap + n_out] = tr[i].score;
                ++n_out;
            }
            tr[w++] = tr[i];
        }
        tr.resize(w);
        for (int i = n_out; i < kCap; ++i)
            out_freq_host[f * kCap + i] = out_score_host[f * kCap + i] = 0.0f;
        out_count_host[f] = n_out;
    }
    return AA_OK;
}

// ---------------------------------------------------------------------- sharding --
int aa_shard_frames(int64_t total_frames, int32_t world, int32_t rank, int64_t *f0, int64_t *f1)
{
    if (!f0 || !f1 || world <= 0 || rank < 0 || rank >= world || total_frames < 0)
        return fail(AA_ERR_ARG, "aa_shard_frames: bad argument");
    // 128-bit intermediate: rank * total can exceed 2^63 only for absurd sizes, but be exact.
    *f0 = (int64_t)(((__int128)total_frames * rank) / world);
    *f1 = (int64_t)(((__int128)total_frames * (rank + 1)) / world);
    return AA_OK;
}

int aa_shard_samples(int64_t f0, int64_t f1, int32_t n_fft, int32_t hop, int64_t *s0, int64_t *s1)
{
    if (!s0 || !s1 || f0 < 0 || f1 < f0 || n_fft <= 0 || hop <= 0)
        return fail(AA_ERR_ARG, "aa_shard_samples: bad argument");
    *s0 = f0 * (int64_t)hop;
    *s1 = (f1 == f0) ? *s0 : (f1 - 1) * (int64_t)hop + n_fft;
    return AA_OK;
}

// --------------------------------------------------------------- streaming front --
struct aa_stream {
    aa_stft_plan *plan = nullptr;  // hop == n_fft over gathered frames
    int n_fft = 0, hop = 0, ring_len = 0;
    std::vector<float> ring;
    int64_t write_pos = 0, read_pos = 0, available = 0, abs_read = 0;
    std::vector<float> gathered;
};

int aa_stream_create(const aa_stft_config *cfg, int32_t ring_len, aa_stream **out_stream)
{
    if (!cfg || !out_stream)
        return fail(AA_ERR_ARG, "aa_stream_create: NULL argument");
    *out_stream = nullptr;
    // The frame loop advances by hop and needs n_fft samples (stft.rs:273, 436-437): hop == 0 would
    // emit one frame for ever, hop > n_fft would drive `available` negative (a usize underflow in
    // the reference).  Both reference instances use hop = n_fft / 4.
    if (cfg->n_fft <= 0 || cfg->hop <= 0 || cfg->hop > cfg->n_fft)
        return fail(AA_ERR_ARG, "aa_stream_create: need 0 < hop (%d) <= n_fft (%d)", cfg->hop,
                    cfg->n_fft);
    if (ring_len == 0)
        ring_len = cfg->n_fft * 4 > 8192 ? cfg->n_fft * 4 : 8192;   // stft.rs:171
    if (ring_len < cfg->n_fft)
        return fail(AA_ERR_ARG, "ring_len %d < n_fft %d", ring_len, cfg->n_fft);
    aa_stream *s = new (std::nothrow) aa_stream();
    if (!s)
        return fail(AA_ERR_NOMEM, "aa_stream_create: out of host memory");
    // The GPU side sees the gathered frames as a gap-free buffer with hop == n_fft.
    aa_stft_config c = *cfg;
    c.hop = cfg->n_fft;
    int rc = aa_stft_plan_create(&c, nullptr, &s->plan);
    if (cfg->window == AA_WINDOW_CUSTOM)
        rc = fail(AA_ERR_UNSUPPORTED, "aa_stream: custom windows are not supported");
    if (rc != AA_OK) {
        aa_stream_destroy(s);
        return rc;
    }
    s->n_fft = cfg->n_fft;
    s->hop = cfg->hop;
    s->ring_len = ring_len;
    s->ring.assign(ring_len, 0.0f);
    *out_stream = s;
    return AA_OK;
}

int aa_stream_destroy(aa_stream *s)
{
    if (!s)
        return AA_OK;
    aa_stft_plan_destroy(s->plan);
    delete s;
    return AA_OK;
}

int aa_stream_push(aa_stream *s, const float *slot_host, int32_t slot_len, float *out_host,
                   int64_t *frame_starts, int64_t max_frames, int64_t *n_emitted)
{
    if (!s || (!slot_host && slot_len > 0) || slot_len < 0 || !out_host || !n_emitted ||
        max_frames < 0)
        return fail(AA_ERR_ARG, "aa_stream_push: bad argument");
    *n_emitted = 0;
    // ingest — stft.rs:247-255 (per-sample modulo write)
    for (int i = 0; i < slot_len; ++i) {
        s->ring[s->write_pos] = slot_host[i];
        s->write_pos = (s->write_pos + 1) % s->ring_len;
        s->available += 1;
    }
    // overflow: drop the oldest samples — stft.rs:262-266
    if (s->available > s->ring_len) {
        int64_t excess = s->available - s->ring_len;
        s->read_pos = (s->read_pos + excess) % s->ring_len;
        s->abs_read += excess;
        s->available = s->ring_len;
    }
    // frame loop — stft.rs:273, 296-299 (gather only; window + FFT run on the GPU), 436-437
    int64_t emitted = 0;
    s->gathered.clear();
    while (s->available >= s->n_fft && emitted < max_frames) {
        if (frame_starts)
            frame_starts[emitted] = s->abs_read;
        size_t base = s->gathered.size();
        s->gathered.resize(base + s->n_fft);
        for (int i = 0; i < s->n_fft; ++i)
            s->gathered[base + i] = s->ring[(s->read_pos + i) % s->ring_len];
        s->read_pos = (s->read_pos + s->hop) % s->ring_len;
        s->abs_read += s->hop;
        s->available -= s->hop;
        ++emitted;
    }
    if (emitted > 0) {
        int rc = aa_stft_execute_host(s->plan, s->gathered.data(), (int64_t)s->gathered.size(),
                                      out_host);
        if (rc != AA_OK)
            return rc;
    }
    *n_emitted = emitted;
    return AA_OK;
}

}  // extern "C"

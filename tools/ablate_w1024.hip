// Diagnostic (never shipped): times the v3 kernel with parts removed (outputs are wrong by
// construction; only the time matters) to find what the real pipeline is bound by.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-strict-aliasing -DAA_ABLATE -I audio_analyzer_rs_amd/csrc -o tools/bin/ablate_w1024 tools/ablate_w1024.hip
#include "aa_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
#include <string>
#include <cstdlib>
#include <ctime>

static float *dx, *dout, *dtab;
static unsigned *dcnt;
static const int hop = 256;
static const int64_t n_samples = 28800000, n_frames = (n_samples - 1024) / hop + 1;

template <int ABL>
double run(int fpw, const char *name, int force_per_cu = 0)
{
    const int64_t chunks = (n_frames + fpw - 1) / fpw;
    const size_t lds = aa::wave1024_lds_bytes(fpw, hop);
    int per_cu = 1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, aa::stft_wave1024_kernel<aa::kMag, ABL, true>, 256, lds);
    if (force_per_cu) per_cu = force_per_cu;
    int grid = 256 * per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL((aa::stft_wave1024_kernel<aa::kMag, ABL, true>), dim3(grid), dim3(256), lds, 0, dx, dout, dtab,
                               (int64_t)0, n_frames, hop, fpw, chunks, 1e-10f, (const float *)nullptr, aa::MelSched(), (const float *)nullptr, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms / 20);
    }
    std::sort(ts.begin(), ts.end());
    std::vector<unsigned long long> clk(4 * 8192);
    hipMemcpyFromSymbol(clk.data(), HIP_SYMBOL(aa::aa_clk_dbg), clk.size() * 8);
    std::vector<double> life, start, endt;
    unsigned long long first = ~0ull;
    int nb = grid < 8192 ? grid : 8192;
    for (int b = 0; b < nb; ++b) first = std::min(first, clk[4 * b + 2]);
    for (int b = 0; b < nb; ++b) {
        life.push_back(clk[4 * b + 1] * 0.01);
        start.push_back((clk[4 * b + 2] - first) * 0.01);
        endt.push_back((clk[4 * b + 2] - first + clk[4 * b + 1]) * 0.01);
    }
    if (const char *dump = getenv("AA_CLK_DUMP")) {      // raw per-workgroup records of the last launch of this run
        std::string path = std::string(dump) + "_" + std::to_string(ABL) + "_" + std::to_string(per_cu) + ".bin";
        if (FILE *f = fopen(path.c_str(), "wb")) { fwrite(clk.data(), 8, 4 * (size_t)nb, f); fclose(f); }
    }
    std::sort(life.begin(), life.end()); std::sort(endt.begin(), endt.end());
    printf("fpw %2d  %-40s %d WG/CU  median %.2f us  min %.2f us | WG life min/med/max %.1f/%.1f/%.1f  end p10/p50/p90/max %.1f/%.1f/%.1f/%.1f\n",
           fpw, name, per_cu, ts[ts.size() / 2] * 1e3, ts[0] * 1e3, life[0], life[nb / 2], life[nb - 1], endt[nb / 10], endt[nb / 2], endt[nb * 9 / 10], endt[nb - 1]);
    return ts[ts.size() / 2];
}

// AA_ABL_SUSTAIN=<seconds>: every variant back to back for that long, the rate measured over the
// last third, with wall-clock marks so that a rocm-smi sampler running beside (tools/power_by_variant.sh)
// can be lined up: which part of the kernel the package power goes to.
template <int ABL, bool WIDE = true>
void sustain(int fpw, const char *name, double seconds)
{
    const int64_t chunks = (n_frames + fpw - 1) / fpw;
    const size_t lds = aa::wave1024_lds_bytes(fpw, hop);
    int per_cu = 1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, aa::stft_wave1024_kernel<aa::kMag, ABL, WIDE>, 256, lds);
    const int grid = 256 * per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const double t_begin = ts.tv_sec + 1e-9 * ts.tv_nsec;
    double now = t_begin, last_us = 0;
    while (now - t_begin < seconds) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 500; ++i)
            hipLaunchKernelGGL((aa::stft_wave1024_kernel<aa::kMag, ABL, WIDE>), dim3(grid), dim3(256), lds, 0, dx, dout, dtab,
                               (int64_t)0, n_frames, hop, fpw, chunks, 1e-10f, (const float *)nullptr, aa::MelSched(), (const float *)nullptr, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        last_us = ms * 2.0;     // per launch
        clock_gettime(CLOCK_REALTIME, &ts);
        now = ts.tv_sec + 1e-9 * ts.tv_nsec;
    }
    printf("SUSTAIN %-44s begin %.2f end %.2f  last %.2f us/launch\n", name, t_begin, now, last_us);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int fpw = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<float> hx(n_samples), win(1024), tab(52 * 64);
    unsigned s = 12345;
    for (auto &v : hx) { s = (s * 1103515245u + 12345u) & 0x7fffffffu; v = (float)s / 1073741824.0f - 1.0f; }
    for (int i = 0; i < 1024; ++i) win[i] = 0.5f - 0.5f * cosf(2.0f * 3.14159265f * i / 1024.0f);
    const double kPi = 3.14159265358979323846;
    for (int lane = 0; lane < 64; ++lane) {
        int n2 = lane, b = lane >> 3;
        for (int j = 0; j < 8; ++j) for (int c = 0; c < 2; ++c) tab[(2 * j + c) * 64 + lane] = 0.5f * win[2 * (64 * j + n2) + c];
        for (int k = 1; k < 8; ++k) {
            double aA = -2.0 * kPi * (n2 * k) / 512.0, aB = -2.0 * kPi * (b * k) / 64.0;
            tab[(16 + 2 * (k - 1)) * 64 + lane] = cos(aA); tab[(17 + 2 * (k - 1)) * 64 + lane] = sin(aA);
            tab[(30 + 2 * (k - 1)) * 64 + lane] = cos(aB); tab[(31 + 2 * (k - 1)) * 64 + lane] = sin(aB);
        }
        for (int j = 0; j < 4; ++j) { double th = 2.0 * kPi * (lane + 64 * j) / 1024.0; tab[(44 + 2 * j) * 64 + lane] = -sin(th); tab[(45 + 2 * j) * 64 + lane] = -cos(th); }
    }
    hipMalloc(&dcnt, 256);
    hipMalloc(&dx, n_samples * 4); hipMalloc(&dout, n_frames * 513 * 4); hipMalloc(&dtab, tab.size() * 4);
    hipMemcpy(dx, hx.data(), n_samples * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 8; ++i) run<0>(fpw, "warm-up (ignore)");
    if (const char *sus = getenv("AA_ABL_SUSTAIN")) {
        const double sec = atof(sus);
        sustain<0>(fpw, "full kernel", sec);
        sustain<1>(fpw, "no global stores", sec);
        sustain<256>(fpw, "stores hit 32 L2-resident rows", sec);
        sustain<8>(fpw, "no DMA after first chunk", sec);
        sustain<2 | 4 | 16>(fpw, "no exchanges at all", sec);
        sustain<1 | 8 | 32>(fpw, "no stores, no DMA, no barrier", sec);
        sustain<1 | 2 | 4 | 8 | 16 | 32>(fpw, "math only (+frame LDS reads)", sec);
        sustain<1024>(fpw, "no priority rotation", sec);
        sustain<0, false>(fpw, "rows stored from registers (9 dword stores)", sec);
        sustain<0>(fpw, "full kernel (again)", sec);
        return 0;
    }
    if (getenv("AA_ABL_ONLY")) {          // just the shipped kernel (per-workgroup clock records via AA_CLK_DUMP)
        for (int i = 0; i < 3; ++i) run<0>(fpw, "full kernel");
        return 0;
    }
    for (int pc : {2, 3, 4, 5}) run<0>(fpw, "full kernel", pc);
    run<1>(fpw, "no global stores");
    run<256>(fpw, "stores hit 32 L2-resident rows");
    run<2>(fpw, "no E2 (register exchange)");
    run<4>(fpw, "no E1 (LDS exchange)");
    run<8>(fpw, "no DMA after first chunk");
    run<16>(fpw, "no split strip");
    run<32>(fpw, "no per-chunk wait+barrier (+races)");
    run<1 | 8 | 32>(fpw, "no stores, no DMA, no barrier");
    run<2 | 4 | 16>(fpw, "no exchanges at all");
    run<1 | 2 | 4 | 8 | 16 | 32>(fpw, "math only (+frame LDS reads)");
    run<512>(fpw, "E2 through LDS instead of registers");
    run<512 | 1>(fpw, "E2 through LDS, no global stores");
    run<1024>(fpw, "no priority rotation (residents finish in dispatch order)");
    run<1024 | 1>(fpw, "no priority rotation, no global stores");
    run<0>(fpw, "full kernel (again)");
    run<1024>(fpw, "no priority rotation (again)");
    run<512>(fpw, "E2 through LDS (again)");
    return 0;
}

// Diagnostic build (never shipped): the v3 kernel with in-kernel s_memtime phase stamps.
// Prints each phase's share of the summed wave time.  Read SHARES, not run time.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DAA_STAMPS -I audio_analyzer_rs_amd/csrc -o tools/bin/stamp_w1024 tools/stamp_w1024.hip
#include "aa_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>

int main(int argc, char **argv)
{
    const int fpw = argc > 1 ? atoi(argv[1]) : 8;
    const int hop = 256;
    const int64_t n_samples = 28800000, n_frames = (n_samples - 1024) / hop + 1;
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
    float *dx, *dout, *dtab; unsigned long long *ddbg;
    hipMalloc(&dx, n_samples * 4); hipMalloc(&dout, n_frames * 513 * 4); hipMalloc(&dtab, tab.size() * 4);
    hipMemcpy(dx, hx.data(), n_samples * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    const int64_t chunks = (n_frames + fpw - 1) / fpw;
    const size_t lds = aa::wave1024_lds_bytes(fpw, hop);
    int per_cu = 1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, aa::stft_wave1024_v3_kernel<aa::kMag, 0, true>, 256, lds);
    int grid = 256 * per_cu;
    hipMalloc(&ddbg, (size_t)grid * 4 * 12 * 8);
    printf("fpw %d lds %zu B, %d WG/CU, grid %d\n", fpw, lds, per_cu, grid);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ddbg, 0, (size_t)grid * 4 * 12 * 8);
        hipLaunchKernelGGL((aa::stft_wave1024_v3_kernel<aa::kMag, 0, true>), dim3(grid), dim3(256), lds, 0, dx, dout, dtab,
                           (int64_t)0, n_frames, hop, fpw, chunks, 1e-10f, ddbg);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h((size_t)grid * 4 * 12);
    hipMemcpy(h.data(), ddbg, h.size() * 8, hipMemcpyDeviceToHost);
    const char *names[12] = {"prologue", "vmcnt(0) wait", "barrier", "DMA issue", "frame read+wait", "window+passA+tw",
                             "E1 LDS xchg", "passB+tw", "E2 regs", "passC", "split LDS xchg", "split+mag+stores"};
    double tot[12] = {0}, all = 0;
    for (size_t w = 0; w < (size_t)grid * 4; ++w) for (int i = 1; i < 12; ++i) { tot[i] += h[w * 12 + i]; all += h[w * 12 + i]; }
    // prologue: first stamp is absolute time, skip
    double frames_per_wave = (double)n_frames / (grid * 4);
    for (int i = 1; i < 12; ++i)
        printf("%2d %-20s %6.2f %%   %8.0f cycles/wave   %7.1f cycles/frame\n", i, names[i], 100.0 * tot[i] / all,
               tot[i] / (grid * 4), tot[i] / (grid * 4) / frames_per_wave);
    printf("total %.0f cycles/wave, %.1f frames/wave, %.0f cycles/frame/wave\n", all / (grid * 4), frames_per_wave, all / (grid * 4) / frames_per_wave);
    return 0;
}

// Build-defined post-ops on the power spectrogram (ABSENT from the reference — SURVEY.md
// §8a rows a10/a11; parity is pinned only against the build's own float64 definition in
// oracle/truth.py):
//   mel     [frames x n_bins] . [n_bins x n_mels]  on the f32 matrix cores
//           (v_mfma_f32_32x32x2_f32: exact f32 fma chain — bf16/f16 inputs would break the
//           1e-5 budget), skipping the K range where a 32-mel column tile is all zero
//           (triangular filters: each tile touches a narrow band of bins);
//   log     10*log10(max(mel, floor));
//   DCT-II  orthonormal, [n_mels -> n_mfcc], one row per workgroup.
// Also the stereo -> mono average of the capture callback (src/audio_io/mod.rs:784-794).
#include "aa_internal.hpp"

namespace aa {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMelKChunk = 64;          // bins staged per LDS round
constexpr int kMelRows = 32;            // frames per workgroup
constexpr int kMelLdA = kMelKChunk + 1; // +1: lanes of one MFMA operand read 32 rows, same k

// power: [n_frames x n_bins] row-major.  melT: [n_bins_pad x n_mels_pad] row-major (filter
// weights transposed, zero padded to a multiple of 32 mels and of kMelKChunk bins).
// tile_k: per 32-mel column tile, [k_lo, k_hi) bin range that holds its non-zeros
// (multiples of kMelKChunk).  out: [n_frames x n_mels].
// Workgroup = 256 threads = 4 waves = 32 frames x 128 mels (one 32x32 tile per wave);
// grid.y walks further groups of 128 mels.
template <bool LOG>
__global__ __launch_bounds__(256) void mel_mfma_kernel(const float *__restrict__ power,
                                                       const float *__restrict__ melT,
                                                       const int2 *__restrict__ tile_k,
                                                       float *__restrict__ out, int64_t n_frames,
                                                       int n_bins, int n_mels, int n_mels_pad,
                                                       float log_floor)
{
    __shared__ float As[kMelRows * kMelLdA];
    __shared__ float Bs[kMelKChunk * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * kMelRows;
    const int col0 = blockIdx.y * 128;
    const int tile = blockIdx.y * 4 + wave;
    // the workgroup walks the union of its four tiles' K ranges
    int k_lo = 1 << 30, k_hi = 0;
    for (int w = 0; w < 4; ++w) {
        if (col0 + 32 * w < n_mels_pad) {
            int2 r = tile_k[blockIdx.y * 4 + w];
            k_lo = r.x < k_lo ? r.x : k_lo;
            k_hi = r.y > k_hi ? r.y : k_hi;
        }
    }
    const bool wave_active = col0 + 32 * wave < n_mels_pad;
    const int2 my = wave_active ? tile_k[tile] : make_int2(0, 0);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ai = lane & 31, ak = lane >> 5;

    for (int kb = k_lo; kb < k_hi; kb += kMelKChunk) {
        // stage A: 32 rows x 64 bins (coalesced along bins)
        for (int e = tid; e < kMelRows * kMelKChunk; e += 256) {
            const int r = e >> 6, k = e & 63;
            const int64_t gr = row0 + r;
            const int gk = kb + k;
            As[r * kMelLdA + k] = (gr < n_frames && gk < n_bins) ? power[gr * n_bins + gk] : 0.0f;
        }
        // stage B: 64 bins x 128 mels
        for (int e = tid; e < kMelKChunk * 128; e += 256) {
            const int k = e >> 7, c = e & 127;
            Bs[e] = (col0 + c < n_mels_pad) ? melT[(int64_t)(kb + k) * n_mels_pad + col0 + c] : 0.0f;
        }
        __syncthreads();
        if (wave_active && kb + kMelKChunk > my.x && kb < my.y) {
#pragma unroll 8
            for (int k = 0; k < kMelKChunk; k += 2) {
                const float a = As[ai * kMelLdA + k + ak];
                const float b = Bs[(k + ak) * 128 + 32 * wave + ai];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (!wave_active)
        return;
    // C/D layout of the 32x32 f32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int col = col0 + 32 * wave + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gr < n_frames && col < n_mels) {
            float v = acc[r];
            if (LOG)
                v = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(v, log_floor));
            out[gr * n_mels + col] = v;
        }
    }
}

// MFCC: out[f][c] = sum_m dct[c][m] * logmel[f][m]; one frame per 64-thread workgroup.
__global__ __launch_bounds__(64) void dct_rows_kernel(const float *__restrict__ logmel,
                                                      const float *__restrict__ dct,
                                                      float *__restrict__ out, int n_mels,
                                                      int n_mfcc)
{
    extern __shared__ float row[];
    const int64_t f = blockIdx.x;
    for (int m = threadIdx.x; m < n_mels; m += 64)
        row[m] = logmel[f * n_mels + m];
    __syncthreads();
    for (int c = threadIdx.x; c < n_mfcc; c += 64) {
        float s = 0.0f;
        for (int m = 0; m < n_mels; ++m)
            s = fmaf(dct[c * n_mels + m], row[m], s);
        out[f * n_mfcc + c] = s;
    }
}

// mixed = (sum of the first min(channels, 2) channels) / that count — mod.rs:784-794
__global__ void downmix_kernel(const float *__restrict__ in, float *__restrict__ out,
                               int64_t n_frames, int channels)
{
    const int use = channels < 2 ? channels : 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_frames;
         i += (int64_t)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int c = 0; c < use; ++c)
            acc += in[i * channels + c];
        out[i] = acc / (float)use;
    }
}

}  // namespace

hipError_t launch_mel(bool take_log, const float *power, const float *melT, const int *tile_k,
                      float *out, int64_t n_frames, int n_bins, int n_mels, int n_mels_pad,
                      float log_floor, hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    dim3 grid((unsigned)((n_frames + kMelRows - 1) / kMelRows), (unsigned)((n_mels_pad + 127) / 128));
    const int2 *tk = reinterpret_cast<const int2 *>(tile_k);
    if (take_log)
        hipLaunchKernelGGL(mel_mfma_kernel<true>, grid, dim3(256), 0, stream, power, melT, tk, out,
                           n_frames, n_bins, n_mels, n_mels_pad, log_floor);
    else
        hipLaunchKernelGGL(mel_mfma_kernel<false>, grid, dim3(256), 0, stream, power, melT, tk, out,
                           n_frames, n_bins, n_mels, n_mels_pad, log_floor);
    return hipGetLastError();
}

hipError_t launch_dct_rows(const float *logmel, const float *dct, float *out, int64_t n_frames,
                           int n_mels, int n_mfcc, hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(dct_rows_kernel, dim3((unsigned)n_frames), dim3(64), sizeof(float) * n_mels,
                       stream, logmel, dct, out, n_mels, n_mfcc);
    return hipGetLastError();
}

hipError_t launch_downmix(const float *in, float *out, int64_t n_frames, int channels,
                          hipStream_t stream)
{
    if (n_frames <= 0)
        return hipSuccess;
    int64_t blocks = (n_frames + 255) / 256;
    if (blocks > 4096)
        blocks = 4096;
    hipLaunchKernelGGL(downmix_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, n_frames,
                       channels);
    return hipGetLastError();
}

}  // namespace aa

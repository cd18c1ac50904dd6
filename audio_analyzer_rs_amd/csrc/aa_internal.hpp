// Internal interface between the C-ABI layer (aa_abi.cpp) and the gfx950 kernels
// (aa_kernels.hip).  Not installed; the public surface is include/aa_stft.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

namespace aa {

// Sets the calling thread's aa_last_error() text and returns `code` (defined in aa_abi.cpp).
int abi_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Selects device `dev` for the lifetime of the guard and restores the previous one: every entry point that takes a
// plan or a communicator acts on ITS device, whatever the calling thread had selected (a new host thread sits on 0).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess))
            ok = true;
    }
    ~DeviceGuard()
    {
        if (ok && prev >= 0)
            (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Experiment knobs exist only in libraries built with -DAA_EXPERIMENTS=1 (tools/: A/B builds, path selectors of the
// profiling scripts): the shipped library never reads the process environment - neither at plan creation nor on a
// launch - except AA_RCCL_PATH, the deployment's location of librccl.so (aa_comm.cpp).  Values that do not parse
// or fall outside [lo, hi] are ignored.
#ifndef AA_EXPERIMENTS
#define AA_EXPERIMENTS 0
#endif
inline int experiment_knob(const char *name, int dflt, int lo, int hi)
{
#if AA_EXPERIMENTS
    if (const char *e = getenv(name)) {
        char *end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end != e && *end == '\0' && v >= lo && v <= hi)
            return (int)v;
    }
#else
    (void)name;
    (void)lo;
    (void)hi;
#endif
    return dflt;
}

// Output kinds handled inside the FFT kernels' epilogue (AA_OUT_COMPLEX..LOGPOWER); kLogMel is the fused
// triangular-filterbank reduction of the 1024 / 2048 wave kernels as a gather on the VALU with the weights rebuilt
// from the triangle edges (the fallback: any band count, any hop, 4-byte-aligned rows).
// kLogMelM / kMfccM (r4; 1024 kernel at 8 frames per chunk, 2048 kernel at hop 512; n_mels % 4 == 0): the mel
// contraction band-blocked on v_mfma_f32_4x4x1_16b_f32 (see "band-blocked mel contraction" below); kMfccM adds the
// DCT-II fused behind it on v_mfma_f32_16x16x4_f32 (the log-mel rows of 16 frames stay in LDS).
// (Rounds 1-3 had kLogMelR / kMfccR = 5 / 6: the gather with resident / streamed weights; replaced by the M kinds.)
enum OutKind : int { kComplex = 0, kMag = 1, kPower = 2, kLogPower = 3, kLogMel = 4, kLogMelM = 7, kMfccM = 8 };

// ---- fused mel reduction (wave kernels, kLogMel) ------------------------------------
// The filterbank is triangular: 98 % zeros, every bin feeds at most two filters, so the fused
// path evaluates it as a gather on the VALU straight from the power row in LDS instead of a
// dense GEMM.  A wave reduces its F frames (2 at 1024, 1 at 2048) with 64/F lanes per frame; a
// group g covers (64/F) >> part_log2[g] consecutive mels from mel_base[g], each filter's support
// cut into 1 << part_log2[g] K-slices of trips[g] bins (narrow low filters: many mels per group,
// wide high filters: more slices).  The weight of a filter at offset i from its first bin is
// med3(U0 + i DU, D0 - i DD, 0) (the two edges of the triangle, Slaney norm folded in), so no
// weight table is needed.
// Per-mel parameters, kMelParDwords floats each: {k_start (int bits), U0, DU, D0, DD, 0}; entry
// kMelMaxMels is all zero (lanes past n_mels read it).
constexpr int kMelMaxMels = 128;
constexpr int kMelMaxGroups = 8;
constexpr int kMelParDwords = 6;
constexpr int kMelParTableDwords = (kMelMaxMels + 1) * kMelParDwords + 2;   // 776, keeps 16-byte size
constexpr int kMelStageDwords = 256;                // per wave: mel values of up to two frames
constexpr int kMelLdsDwords = kMelParTableDwords + 4 * kMelStageDwords;
// kMfccM: log-mel rows of two chunks (16 frames) in LDS, row stride 136 floats and the columns of rows 4..7 / 12..15
// swizzled (col ^ 4): the B-operand reads of the DCT's MFMA - 16 frames x 4 K-groups, 16 bytes each - and the row writes
// are then conflict free on the lane groups the hardware serves ds_read_b128 / ds_write_b128 in (r4; 132 floats before,
// chosen for consecutive 16-lane groups, which they are not); the fold about the middle of the mel axis
// (u = L[m] + L[M-1-m] for the even coefficients, v = L[m] - L[M-1-m] for the odd ones: the DCT-II matrix is symmetric /
// antisymmetric there) happens in the B operand.  The DCT matrix comes as a per-lane table [tile][4][64] float4:
// tile = (parity, half) = 16 coefficients c = 2 (16 half + i) + parity, K ordered m = 16 S + 4 (lane / 16) + t: 4 tiles =
// up to 64 coefficients, n_mels % 4 == 0 and <= 128.
constexpr int kMfccRowStride = 136;   // (B-operand reads conflict free on the hardware lane groups of ds_read_b128; r2-r3: 132)
constexpr int kMfccLdsDwords = 16 * kMfccRowStride;
constexpr int kMfccTiles = 4;
// the coefficients of a 16-frame block leave through an LDS tile [16][n_mfcc] (the block's rows are one contiguous span of
// the output: coalesced stores by the whole workgroup instead of four scattered dwords per lane)
constexpr int kMfccOutTileDwords = 16 * (16 * kMfccTiles + 1) + 12;      // row stride n_mfcc | 1 (bank spread); a multiple of 4 dwords
constexpr int kMfccTableDwords = kMfccTiles * 4 * 64 * 4;
struct MelSched {
    int n_mels = 0;
    int n_groups = 0;
    int trips[kMelMaxGroups] = {0, 0, 0, 0, 0, 0, 0, 0};       // bins per K-slice (multiples of 4)
    int mel_base[kMelMaxGroups] = {0, 0, 0, 0, 0, 0, 0, 0};
    int part_log2[kMelMaxGroups] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0, 1 or 2
    // kLogMelM / kMfccM: steps of the band-blocked contraction (the groups above are unused there)
    int n_steps = 0;
};

// ---- band-blocked mel contraction on the matrix cores (kLogMelM / kMfccM, r4) -------------------
// After a workgroup barrier the power rows of all 8 frames of a chunk are visible in the four waves'
// scratch areas, and every wave contracts a QUARTER OF THE FILTERBANK with all 8 frames instead of
// the whole filterbank with its own 2: v_mfma_f32_4x4x1_16b_f32 is sixteen independent 4 x 4 x 1
// products (block = lane / 4; A: i = lane % 4, B: j = lane % 4, D[register i][j]; measured 8.9 cycles
// with four accumulators, 9.8 with two: tools/ubench_mfma4x4.hip), here block = a quad of 4
// consecutive mels (i) x 4 frames (j) x one bin (k), each block walking ITS OWN bin range - the band
// structure of the triangular filterbank (2 of the 4 mels of a quad are non-zero at any bin, against
// 2 of 128 for a dense GEMM).  A step is 4 bins: one ds_read_b128 of the power row (B operand, 4
// MFMAs' worth) and one resident float4 of weights (A operand) per lane; frames j and 4 + j share the
// A operand (two accumulator sets).  A quad of more than kMelMfmaSteps pieces is cut into K-slices on
// consecutive blocks of the wave; every block leaves its 4 x 8 partial sums in LDS (the chunk's sample
// buffer is dead by then) and the wave's lanes, re-labelled (quad slot, frame), add the slices of
// their quad, take the logs and write the 16-byte piece of the row: all 64 lanes busy for 8 quads.
// Per-wave table: [kMelMfmaSteps][64][4] weights, then [2][64] dwords: the byte offset of the lane's
// first piece in a power row, and the emission word of the lane as (quad slot = lane / 8, frame =
// lane % 8): bits 0-4 slices of the quad (0: idle lane), 5-8 its first block, 9-13 the quad, 14-15 the shift of
// the second partial-sum set (below; the same in every lane), 24-28 the wave's largest slice count (the same in
// every lane).
constexpr int kMelMfmaSteps = 4;
constexpr int kMelMfmaRowB = 528;                   // float offset of the wave's second power row in its scratch
constexpr int kMelMfmaScratchDwords = 1072;         // scratch stride of these kinds: 268 float4 = 12 mod 16 -> the four
                                                    // frames a block reads sit in four different bank quarters
constexpr int kMelMfmaWaveDwords = kMelMfmaSteps * 64 * 4 + 2 * 64;
constexpr int kMelMfmaTableDwords = 4 * kMelMfmaWaveDwords;
// partial sums of a workgroup: [wave][frame set][lane][4].  The second frame set (frames 4..7) starts 64 + 4 d float4
// behind the first, d = 0..3 from bits 14-15 of the wave's emission words: the shift that spreads a slot's two sets
// over the 16-byte bank slots of a ds_read_b128 lane group for the wave's mix of slice counts (build_mel_mfma)
constexpr int kMelMfmaPartWaveDwords = (64 + 12 + 64) * 4;
constexpr int kMelMfmaPartDwords = 4 * kMelMfmaPartWaveDwords;
// the 2048 kernel (hop 512, 4 frames per chunk = the 4 columns of a block: one frame set, 8 emission slots of 4 frames x
// the two halves of a quad per wave; up to 7 steps of resident weights; same per-wave table layout)
constexpr int kMelMfmaSteps2048 = 7;
constexpr int kMelMfmaWaveDwords2048 = kMelMfmaSteps2048 * 64 * 4 + 2 * 64;
constexpr int kMelMfmaTableDwords2048 = 4 * kMelMfmaWaveDwords2048;

// ---- wave-per-frame 1024-point kernel --------------------------------------
// Per-lane constant table, row-major [kWave1024Rows][64] floats (see
// build_wave1024_table in aa_abi.cpp for the row meaning).
constexpr int kWave1024Rows = 52;
constexpr int kWave1024Threads = 256;              // 4 waves per workgroup
constexpr int kWave1024ScratchDwords = 1056;       // per-wave E1 scratch / split strip / row staging
// LDS dwords a workgroup needs for `frames_per_wg` frames: double-buffered unpadded staging
// of (frames_per_wg-1)*hop + 1024 samples + the four per-wave scratch areas.
// mel: 0 none, 1 fused mel reduction (kLogMel), 3 kLogMelM, 4 kMfccM
inline int64_t wave1024_lds_dwords(int frames_per_wg, int hop, int mel = 0)
{
    const int64_t span = (int64_t)(frames_per_wg - 1) * hop + 1024;
    // (the partial sums of the mel contraction live in the chunk's dead sample buffer when it is large enough)
    if (mel >= 3)
        return 2 * span + 4 * kMelMfmaScratchDwords + (mel == 4 ? kMfccLdsDwords + kMfccOutTileDwords : 0) + 4 +
               (span < kMelMfmaPartDwords ? kMelMfmaPartDwords : 0);       // (+ 4: the zero slot of the slice sums)
    return 2 * span + 4 * kWave1024ScratchDwords + (mel ? kMelLdsDwords : 0);
}
inline size_t wave1024_lds_bytes(int frames_per_wg, int hop, int mel = 0)
{
    return sizeof(float) * (size_t)wave1024_lds_dwords(frames_per_wg, hop, mel);
}

// Several equal-length channels in one launch (aa_stft_execute_channels_device): channel c reads
// x + c in_stride and writes out + c out_stride (floats, multiples of 4); n = 1: a plain launch.
struct Channels {
    int n = 1;
    int64_t in_stride = 0, out_stride = 0;
};
// mel_par / sched: kLogMel (the parameter table and the groups), kLogMelM / kMfccM (the kMelMfmaTableDwords table, n_mels
// and n_steps); dct_tab / n_mfcc: kMfccM only (kMfccTableDwords floats; n_mfcc <= 16 kMfccTiles).
hipError_t launch_wave1024(int out_kind, const float *x, float *out, const float *table,
                           int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                           float log_floor, hipStream_t stream, const float *mel_par = nullptr,
                           const MelSched *sched = nullptr, const float *dct_tab = nullptr, int n_mfcc = 0,
                           const Channels *channels = nullptr);

// kMfccM of launch_wave1024, defined in aa_kernels_mfcc.hip (arguments already validated by launch_wave1024)
hipError_t launch_wave1024_mfcc(const float *x, float *out, const float *table, int64_t frame_begin, int64_t n_frames, int hop,
                                float log_floor, hipStream_t stream, const float *mel_par, const MelSched &sch,
                                const float *dct_tab, int n_mfcc, const Channels &ch);

// ---- wave-per-frame 2048-point kernel (tuner geometry) ------------------------------
constexpr int kWave2048Rows = 92;
// behind the rows, as float4 per lane: [8][64] window quadruples, [4][64] pairs of combine twiddles, [4][64] pairs
// of split twiddles (stft_wave2048_mel_kernel streams these two per frame), [4][64] + [4][64] pairs of pass-A /
// pass-B twiddles
constexpr int kWave2048TableDwords = kWave2048Rows * 64 + 3 * 8 * 64 * 4;
inline int64_t wave2048_lds_dwords(int frames_per_wg, int hop, bool mel = false)
{
    const int64_t span = (int64_t)(frames_per_wg - 1) * hop + 2048;
    return 2 * span + 4 * kWave1024ScratchDwords + (mel ? kMelLdsDwords : 0);
}
inline size_t wave2048_lds_bytes(int frames_per_wg, int hop, bool mel = false)
{
    return sizeof(float) * (size_t)wave2048_lds_dwords(frames_per_wg, hop, mel);
}
// Ring variants (hop 512, 4 frames per chunk): `blocks` blocks of 256 floats + the four scratch areas
// [+ the parameter table and the stage (kLogMel: 1)].
inline size_t wave2048_ring_lds_bytes(int blocks, int mel_kind)
{
    return sizeof(float) * (size_t)(blocks * 256 + 4 * kWave1024ScratchDwords +
                                    (mel_kind == 1 ? kMelLdsDwords : 0));
}
// stft_wave2048_mel_kernel: the ring, four scratch areas of the mel kinds' stride, a zero float4, [4][64] float4 of partial sums
inline size_t wave2048_mel_lds_bytes()
{
    return sizeof(float) * (size_t)(30 * 256 + 4 * kMelMfmaScratchDwords + 4 + 4 * 64 * 4);
}
// kLogMelM (hop 512, 4 frames per chunk, 16-byte-aligned out, n_mels % 4 == 0): mel_par is the kMelMfmaTableDwords2048 table,
// sched->n_steps the number of steps.
hipError_t launch_wave2048(int out_kind, const float *x, float *out, const float *table,
                           int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                           float log_floor, hipStream_t stream, const float *mel_par = nullptr,
                           const MelSched *sched = nullptr, const Channels *channels = nullptr);

// ---- several-frames-per-wave kernel for n_fft 256 (onset geometry) and 512 ------------------
constexpr int kWave256Rows = 50;
constexpr int kWave256ScratchDwords = 1152;         // per wave: exchange (1056) / strips / row image
inline int64_t wave_small_lds_dwords(int n_fft, int frames_per_wg, int hop)
{
    const int64_t span = (int64_t)(frames_per_wg - 1) * hop + n_fft;
    return 2 * span + 4 * kWave256ScratchDwords;
}
inline size_t wave_small_lds_bytes(int n_fft, int frames_per_wg, int hop)
{
    return sizeof(float) * (size_t)wave_small_lds_dwords(n_fft, frames_per_wg, hop);
}
hipError_t launch_wave_small(int n_fft, int out_kind, const float *x, float *out, const float *table,
                             int64_t frame_begin, int64_t n_frames, int hop, int frames_per_wg,
                             float log_floor, hipStream_t stream);

// ---- generic kernels (any even length) ---------------------------------------
// Frame t reads x[t*hop + i] * window[i] (window may be null).  Rows of
// out: kComplex -> 2*(n/2+1) floats, else n/2+1 floats.
// pow2 == true : LDS Stockham radix-2;  false: direct DFT from a full twiddle table.
//   tw_m   : W_m^q  (re,im interleaved), q < m/2 (pow2) or q < m (dft), m = n/2
//   tw_split: W_n^k (re,im interleaved), k <= m/2
hipError_t launch_generic_forward(bool pow2, int out_kind, const float *x, float *out,
                                  const float *window, const float *tw_m, const float *tw_split,
                                  int n, int64_t frame_begin, int64_t n_frames, int64_t hop,
                                  float log_floor, hipStream_t stream);
// Inverse C2R, unnormalised: rows of 2*(n/2+1) floats -> rows of n floats.
hipError_t launch_generic_inverse(bool pow2, const float *spec, float *out, const float *tw_m,
                                  const float *tw_split, int n, int64_t batch, hipStream_t stream);

// ---- build-defined post-ops (aa_mel.hip) ----------------------------------------
// C[n_rows x n_cols] = op(A[n_rows x K] . B): B is [k_pad x n_pad] row-major, zero padded
// (k_pad % 32 == 0, n_pad % 32 == 0, n_pad <= 128); tile_k: int pairs [k_lo, k_hi) per 32-column
// tile (multiples of 32); op = 10*log10(max(., floor)) when take_log.
hipError_t launch_rows_gemm(bool take_log, const float *A, int lda, const float *B, const int *tile_k,
                            float *C, int64_t n_rows, int K, int n_cols, int n_pad, float log_floor,
                            hipStream_t stream);
// The three recurrence / per-frame post-ops take an optional batch description: frame_offsets
// [n_streams + 1] (device) of streams concatenated along the frame axis; null = one stream.
hipError_t launch_onset_flux(const float *mags, const float *prev0, float *flux, float *energy,
                             int64_t n_frames, int n_bins, hipStream_t stream,
                             const int64_t *offsets = nullptr, int n_streams = 1);
hipError_t launch_onset_gates(const float *mags, const float *flux, const float *energy,
                              const float *global_floor, float *state, int initialized, int64_t n_frames,
                              int n_bins, float *out_flux, float *out_ema, float *out_excess,
                              int32_t *out_count, uint8_t *out_flags, float *out_velocity, hipStream_t stream,
                              const int64_t *offsets = nullptr, const int32_t *init_flags = nullptr,
                              int n_streams = 1, const uint8_t *suppress = nullptr, int calibrated = 1);
hipError_t launch_tuner_pitches(const float *mags, const float *eff_floor, int64_t n_frames, int n_bins,
                                float bin_width, float min_freq, float max_freq, float *out_freq,
                                float *out_score, int32_t *out_count, hipStream_t stream);
hipError_t launch_tuner_floor(const float *mags, const float *global_floor, float *state,
                              int initialized, float *eff, int64_t n_frames, int n_bins,
                              hipStream_t stream, const int64_t *offsets = nullptr,
                              const int32_t *init_flags = nullptr, int n_streams = 1);
hipError_t launch_downmix(const float *in, float *out, int64_t n_frames, int channels,
                          hipStream_t stream);
// diagnostics: the flux kernel's x / 3 against the IEEE division over `n` consecutive bit patterns; a plain /
// twice-stored nontemporal copy of n_floats floats (n_floats % 4 == 0, 16-byte aligned buffers)
hipError_t launch_flux_div3_check(uint32_t first, uint64_t n, unsigned long long *mismatches_dev, hipStream_t stream);
hipError_t launch_stream_copy(const float *src, float *dst, int64_t n_floats, int writes, hipStream_t stream);

}  // namespace aa

// the HIP device a communicator was created on (aa_comm.cpp; internal: aa_stft_execute_sharded checks it against its plan's)
struct aa_comm;
int aa_comm_device_index(const aa_comm *comm);
